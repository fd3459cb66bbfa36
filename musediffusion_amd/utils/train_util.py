"""`TrainLoop` with the reference's constructor and loop cadence (MuseDiffusion/utils/train_util.py:34-372), so that
run/train.py:132-151 reaches the libmusehip training path without edits.  It is a shell: the step itself is
`train_step.TrainStep` (forward_backward / forward_only / optimize over the kernel-level tape and the fused AdamW + EMA launch),
files are `checkpoint.py` (the reference's names and keys).  What the reference does through `logger` / `dist_util` / `blobfile`
(out of scope) is reduced to what the loop needs: key-value means kept in `self.kvs` (and handed to an optional
`log_fn`), rank / world size from torch.distributed, plain files.
"""
import os

import torch
import torch.distributed as dist

from .. import checkpoint as ckpt
from ..train_step import TrainStep


def update_ema(target_params, source_params, rate=0.99):
    """train_util.py:21-31, for callers that keep EMA copies themselves (TrainLoop's live inside the fused optimizer)."""
    for trg, src in zip(target_params, source_params):
        trg.detach().mul_(rate).add_(src, alpha=1 - rate)


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class TrainLoop:
    """Keyword-only constructor of train_util.py:35-56.  Extras (keyword-only, all optional): `optimizer` (anything with
    grad_norm() / step(lr=); default the fused AdamW + EMA kernel), `log_fn(dict)` called where the reference calls
    `logger.dumpkvs()`, `ddp_kwargs` merged over the reference's DistributedDataParallel arguments (:109-116)."""

    def __init__(self, *, model, diffusion, data, batch_size, microbatch, lr, ema_rate, log_interval, save_interval,
                 resume_checkpoint, schedule_sampler=None, weight_decay=0.0, learning_steps=0, checkpoint_path='',
                 gradient_clipping=-1., eval_data=None, eval_interval=-1, eval_callbacks=(), optimizer=None, log_fn=None,
                 ddp_kwargs=None):
        self.model, self.diffusion = model, diffusion
        self.data, self.eval_data = data, eval_data
        self.batch_size = batch_size
        self.microbatch = microbatch if microbatch > 0 else batch_size
        self.log_interval, self.eval_interval, self.save_interval = log_interval, eval_interval, save_interval
        self.resume_checkpoint = resume_checkpoint
        self.checkpoint_path = checkpoint_path
        self.eval_callbacks = list(eval_callbacks)
        self.global_batch = self.batch_size * _world()
        self.log_fn = log_fn
        self.kvs, self._kv_n = {}, {}

        # :86-87, :121-130 - resume: newest model*.pt of the checkpoint directory, else the file named on the command line
        self.resume_step = 0
        main = self._main_checkpoint()
        if main:
            self.resume_step = self.parse_resume_step_from_filename(main)
            if _rank() == 0:
                self.model.load_state_dict(torch.load(main, map_location=next(self.model.parameters()).device))
            self._sync_params(self.model.parameters())

        if dist.is_available() and dist.is_initialized():                       # :106-116
            from torch.nn.parallel.distributed import DistributedDataParallel
            dev = next(self.model.parameters()).device
            kw = dict(broadcast_buffers=False, bucket_cap_mb=128, find_unused_parameters=False)
            if dev.type == "cuda":
                kw.update(device_ids=[dev], output_device=dev)
            kw.update(ddp_kwargs or {})
            self.use_ddp, self.ddp_model = True, DistributedDataParallel(self.model, **kw)
        else:
            self.use_ddp, self.ddp_model = False, self.model

        self._ts = TrainStep(model, diffusion, microbatch=self.microbatch, lr=lr, weight_decay=weight_decay, ema_rate=ema_rate,
                             learning_steps=learning_steps, gradient_clipping=gradient_clipping, schedule_sampler=schedule_sampler,
                             ddp_model=self.ddp_model if self.use_ddp else None, resume_step=self.resume_step, optimizer=optimizer)
        self.lr, self.ema_rate = self._ts.lr, self._ts.ema_rate
        self.weight_decay, self.learning_steps, self.gradient_clipping = weight_decay, learning_steps, gradient_clipping
        self.schedule_sampler = self._ts.schedule_sampler
        self.model_params = self.master_params = self._ts.model_params
        self.opt = self._ts.opt
        if self.resume_step:                                                     # :93-101 optimizer state and EMA copies of the resumed step
            self._load_optimizer_state(main)
            self._load_ema_parameters(main)
        if torch.cuda.is_available():
            torch.cuda.empty_cache()                                             # :121-123

    # ------------------------------------------------------------------ resume helpers (:121-160)
    def _main_checkpoint(self):
        return self.find_resume_checkpoint(self.checkpoint_path) or self.resume_checkpoint

    @staticmethod
    def _sync_params(params):
        if dist.is_available() and dist.is_initialized():                        # dist_util.sync_params, :141-152
            for p in params:
                dist.broadcast(p.data if isinstance(p, torch.nn.Parameter) else p, 0)

    def _load_optimizer_state(self, main):
        path = self.find_opt_checkpoint(main, self.resume_step)
        if path and hasattr(self.opt, "load_state_dict"):
            self.opt.load_state_dict(torch.load(path, map_location="cpu"))

    def _load_ema_parameters(self, main):
        """:132-147 - the EMA copy of every rate starts from the resumed step's `ema_{rate}_{step}.pt`, or from a copy of the (loaded)
        parameters when that file is missing; rank 0 reads, everyone receives."""
        ema = getattr(self.opt, "ema", None)
        if ema is None:
            return
        names = [n for n, _ in self.model.named_parameters()]
        for i, rate in enumerate(self.ema_rate):
            path = self.find_ema_checkpoint(main, self.resume_step, rate)
            if path and _rank() == 0:
                sd = torch.load(path, map_location="cpu")
                for j, n in enumerate(names):
                    ema[i][j].copy_(sd[n])
            self._sync_params(ema[i])

    # ------------------------------------------------------------------ the loop (:162-186)
    @property
    def step(self):
        return self._ts.step

    @step.setter
    def step(self, v):
        self._ts.step = v

    def run_loop(self):
        while not self.learning_steps or self.step + self.resume_step < self.learning_steps:
            cond = next(self.data)
            self.forward_backward(cond)
            self.optimize()
            self.log_step()
            if self.step % self.log_interval == 0:
                self.dumpkvs()
            if self.eval_data is not None and self.step % self.eval_interval == 0:
                cond_eval = next(self.eval_data)
                self.forward_only(cond_eval)
                for callback in self.eval_callbacks:
                    callback(self)
                self.dumpkvs()
            if self.step > 0 and self.step % self.save_interval == 0:
                self.save()
            self.step += 1
        if (self.step - 1) % self.save_interval != 0:
            self.save()

    __call__ = run_loop

    def forward_backward(self, cond):
        self._log_losses(self._ts.forward_backward(cond))

    def forward_only(self, cond):
        self._log_losses(self._ts.forward_only(cond))

    def zero_grad(self):
        self._ts.zero_grad()

    def optimize(self):
        self.logkv_mean("grad_norm", self._ts.optimize())

    def grad_clip(self):
        if hasattr(self.opt, "clip_grad_norm"):
            self.opt.clip_grad_norm(self.gradient_clipping)
        else:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.gradient_clipping)

    # ------------------------------------------------------------------ logging (:282-292) without a host sync per value
    def _log_losses(self, losses):
        for k, v in losses.items():
            self.logkv_mean(k, v)

    def logkv_mean(self, key, value):
        """running mean like logger.logkv_mean; device scalars stay on the device until dumpkvs"""
        n = self._kv_n.get(key, 0)
        self.kvs[key] = value if n == 0 else (self.kvs[key] * n + value) / (n + 1)
        self._kv_n[key] = n + 1

    def log_step(self):
        self.kvs["step"] = self.step + self.resume_step
        self.kvs["samples"] = (self.step + self.resume_step + 1) * self.global_batch
        self._kv_n.pop("step", None); self._kv_n.pop("samples", None)

    def dumpkvs(self):
        out = {k: (float(v) if torch.is_tensor(v) else v) for k, v in self.kvs.items()}
        self.kvs, self._kv_n = {}, {}
        if self.log_fn is not None:
            self.log_fn(out)
        return out

    # ------------------------------------------------------------------ files (:294-371)
    def save(self):
        if _rank() == 0:
            ckpt.save(self.checkpoint_path, self.step + self.resume_step, self.model, self.opt, self.ema_rate)
        if dist.is_available() and dist.is_initialized():
            dist.barrier()

    @staticmethod
    def parse_resume_step_from_filename(filename):
        """path/to/modelNNNNNN.pt -> NNNNNN (:335-343, same assertion)"""
        filename = os.path.basename(filename)
        assert filename.startswith('model') and filename[-3:] == '.pt', "Invalid model name"
        return int(filename[-9:-3])

    @staticmethod
    def find_resume_checkpoint(log_dir=None):
        """:345-351 looks in the logger's directory, which run/train.py:52 sets to the checkpoint path"""
        return ckpt.find_resume_checkpoint(log_dir) if log_dir else None

    find_ema_checkpoint = staticmethod(ckpt.find_ema_checkpoint)

    @staticmethod
    def find_opt_checkpoint(main_checkpoint, step):
        if not main_checkpoint:
            return None
        path = os.path.join(os.path.dirname(main_checkpoint), f"opt_{step:06d}.pt")
        return path if os.path.exists(path) else None
