"""Factory of the hot path's two objects and the pretrained-weight helpers run/train.py:93-100 calls before TrainLoop:
API of MuseDiffusion/utils/initialization.py:11-136."""
import random

import numpy as np
import torch


def seed_all(seed, deterministic=False, rank=0):
    """utils/initialization.py:11-26 (rank offset unless deterministic)."""
    seed = hash(seed) if deterministic else hash(seed) + rank
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)


def _load_state_dict(path, **kwargs):
    """utils/dist_util.py:118-138 `load_state_dict`: every rank reads the file itself (`torch.load`, CPU by default)."""
    kwargs.setdefault("map_location", "cpu")
    return torch.load(path, **kwargs)


def _barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def _log(msg):
    import logging
    logging.getLogger("musediffusion_amd").info(msg)


def fetch_pretrained_embedding(args):
    """utils/initialization.py:29-51: the `weight` entry of the file `args.pretrained_embedding` names (an `nn.Embedding` state_dict),
    or None.  A width that differs from `args.hidden_dim` OVERWRITES args.hidden_dim (the reference warns and does the same);
    `--freeze_embedding` without a pretrained embedding is the reference's `argparse.ArgumentTypeError`."""
    import os
    if args.pretrained_embedding:
        emb_weight = _load_state_dict(args.pretrained_embedding)["weight"]
        _, orig_hidden_dim = emb_weight.shape
        if orig_hidden_dim != args.hidden_dim:
            import warnings
            warnings.warn("Pretrained embedding %s's hidden_dim %d differs from config's hidden dim %d: args.hidden_dim is overwritten"
                          % (os.path.basename(args.pretrained_embedding), orig_hidden_dim, args.hidden_dim))
            args.hidden_dim = orig_hidden_dim
        return emb_weight
    if getattr(args, "freeze_embedding", False):
        import argparse
        raise argparse.ArgumentTypeError("Cannot turn --freeze_embedding on without --pretrained_embedding!")
    return None


def overload_embedding(model, emb_weight, freeze_embedding):
    """utils/initialization.py:54-68.  The embedding's Parameter is REPLACED (not copied into), so `lm_head.weight` - tied to the old
    Parameter at construction (network.py:56-58) - keeps the old tensor and trains on as a separate parameter, exactly as in the
    reference; `freeze_embedding` then takes only the new embedding out of the optimizer's reach."""
    orig_vocab_size, _ = emb_weight.shape
    assert model.word_embedding.weight.shape[0] == orig_vocab_size
    old = model.word_embedding.weight
    with torch.no_grad():
        model.word_embedding.weight = torch.nn.Parameter(emb_weight.to(device=old.device, dtype=old.dtype))
    if freeze_embedding:
        model.word_embedding.requires_grad_(False)
    _log("### Successfully overloaded pretrained embedding weight.")
    _barrier()
    return model


def fetch_pretrained_denoiser(args):
    """utils/initialization.py:71-76: the state_dict in `args.pretrained_denoiser`, or None."""
    if args.pretrained_denoiser:
        return _load_state_dict(args.pretrained_denoiser)
    return None


def overload_denoiser(model, denoiser_state_dict):
    """utils/initialization.py:79-87: entries of `denoiser_state_dict` whose keys the model has replace the model's; the rest of the
    model keeps its values."""
    model_dict = model.state_dict()
    model_dict.update({k: v for k, v in denoiser_state_dict.items() if k in model_dict})
    model.load_state_dict(model_dict)
    _log("### Successfully overloaded pretrained denoiser dict.")
    _barrier()
    return model


def get_latest_model_path(base_path):
    """utils/initialization.py:90-105: the newest `.pt` file (by mtime) inside the newest sub-directory (by mtime) of `base_path`;
    None when either level is empty or unreadable."""
    import os

    def newest(folder, keep):
        with os.scandir(folder) as entries:
            return max((e.path for e in entries if keep(e)), key=os.path.getmtime, default=None)

    try:
        run_dir = newest(base_path, lambda e: e.is_dir())
        return None if run_dir is None else newest(run_dir, lambda e: e.is_file() and e.name.endswith(".pt"))
    except OSError:
        return None


def create_model_and_diffusion(args, **model_overrides):
    """Build (TransformerNetModel, SpacedDiffusion) from a settings object with the reference's
    field names (hidden_dim, hidden_t_dim, vocab_size, seq_len, dropout, noise_schedule,
    diffusion_steps, timestep_respacing, rescale_timesteps, predict_xstart).

    Optional fields / overrides the reference does not have: bert_hidden, bert_layers, bert_heads,
    bert_ffn (Transformer shape; default bert-base-uncased like network.py:44) and compute_dtype."""
    from ..models.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    from ..models.network import TransformerNetModel

    extra = {}
    for k in ("bert_hidden", "bert_layers", "bert_heads", "bert_ffn", "bert_hidden_dropout", "bert_attention_dropout", "compute_dtype"):
        if k in model_overrides:
            extra[k] = model_overrides[k]
        elif getattr(args, k, None) is not None:
            extra[k] = getattr(args, k)
    model = TransformerNetModel(input_dims=args.hidden_dim, output_dims=args.hidden_dim,
                                hidden_t_dim=args.hidden_t_dim, vocab_size=args.vocab_size, seq_len=args.seq_len,
                                dropout=args.dropout, **extra)
    betas = get_named_beta_schedule(args.noise_schedule, args.diffusion_steps)
    timestep_respacing = args.timestep_respacing or [args.diffusion_steps]
    diffusion = SpacedDiffusion(use_timesteps=space_timesteps(args.diffusion_steps, timestep_respacing), betas=betas,
                                rescale_timesteps=args.rescale_timesteps, predict_xstart=args.predict_xstart)
    return model, diffusion
