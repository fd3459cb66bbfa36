"""Checkpoint I/O in the reference's on-disk format: `model_{step:06d}.pt`,
`ema_{rate}_{step:06d}.pt`, `opt_{step:06d}.pt` holding plain `state_dict`s with the reference's key names
(utils/train_util.py:294-319), newest-checkpoint discovery and step parsing (:336-351), and resume."""
import os
import re

import torch


def parse_resume_step_from_filename(filename):
    """`path/to/model_NNNNNN.pt` -> NNNNNN, 0 when the name does not match (train_util.py:336-343)."""
    m = re.search(r"model_?(\d+)\.pt$", os.path.basename(filename))
    return int(m.group(1)) if m else 0


def find_resume_checkpoint(directory):
    """Last `model*.pt` of `directory` in sorted-name order, or None (train_util.py:346-351: `sorted(...)[-1]`;
    with the zero-padded step in the name that is the highest step, whatever the files' modification times)."""
    try:
        names = sorted(f for f in os.listdir(directory) if f.endswith(".pt") and f.startswith("model"))
    except OSError:
        return None
    return os.path.join(directory, names[-1]) if names else None


def find_ema_checkpoint(main_checkpoint, step, rate):
    """Sibling `ema_{rate}_{step:06d}.pt` of a model checkpoint, or None (train_util.py:354-360)."""
    if main_checkpoint is None:
        return None
    path = os.path.join(os.path.dirname(main_checkpoint), f"ema_{rate}_{step:06d}.pt")
    return path if os.path.exists(path) else None


def save(directory, step, model, optimizer=None, ema_rates=()):
    """Rank-0 save of model / EMA copies / optimizer state (train_util.py:294-319).  Tensors are moved to the host;
    the files load with `torch.load` + `load_state_dict` in the reference unchanged."""
    os.makedirs(directory, exist_ok=True)
    cpu = lambda sd: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}
    torch.save(cpu(model.state_dict()), os.path.join(directory, f"model_{step:06d}.pt"))
    if optimizer is None:
        return
    if ema_rates:
        if not hasattr(optimizer, "ema_state_dict"):
            # the reference always writes its EMA copies (train_util.py:300-308): an optimizer that keeps none cannot stand in for
            # them, and a resume would silently restart the averages from the live weights
            import warnings
            warnings.warn("checkpoint.save: ema_rates %s are configured but %s keeps no EMA copies (no ema_state_dict): no ema_*.pt "
                          "is written and a resume restarts the averages from the live weights" % (list(ema_rates), type(optimizer).__name__))
        else:
            for i, rate in enumerate(ema_rates):
                torch.save(cpu(optimizer.ema_state_dict(i, model)), os.path.join(directory, f"ema_{rate}_{step:06d}.pt"))
    if hasattr(optimizer, "state_dict"):         # opt_{step}.pt whenever there is optimizer state (train_util.py:310-315)
        osd = optimizer.state_dict()
        osd["state"] = {k: cpu(v) for k, v in osd["state"].items()}
        torch.save(osd, os.path.join(directory, f"opt_{step:06d}.pt"))


def resume(directory, model, optimizer=None, ema_rates=(), map_location="cpu"):
    """Load the newest checkpoint of `directory` into model (+ optimizer / EMA copies); returns the step (0: none)."""
    main = find_resume_checkpoint(directory)
    if main is None:
        return 0
    step = parse_resume_step_from_filename(main)
    model.load_state_dict(torch.load(main, map_location=map_location))
    if optimizer is not None:
        opt_path = os.path.join(directory, f"opt_{step:06d}.pt")
        if os.path.exists(opt_path):
            optimizer.load_state_dict(torch.load(opt_path, map_location=map_location))
        names = [n for n, _ in model.named_parameters()]
        for i, rate in enumerate(ema_rates):
            path = find_ema_checkpoint(main, step, rate)
            # no EMA file: the reference starts the EMA from a copy of the loaded parameters (train_util.py:139-152)
            sd = torch.load(path, map_location=map_location) if path else model.state_dict()
            for j, n in enumerate(names):
                optimizer.ema[i][j].copy_(sd[n])
    return step
