"""One optimizer step of the reference's TrainLoop, restated around the libmusehip training path
(MuseDiffusion/utils/train_util.py:188-272): `forward_backward` walks the batch in micro-batches, keeps DDP from
all-reducing on all but the last one (`no_sync`, :212-216), updates a loss-aware sampler per micro-batch (:220-221),
back-propagates `(loss * weights).mean()` of every micro-batch (:222-223; the reference does NOT divide by the
number of micro-batches - gradients add up), then `optimize` clips (optional), anneals the learning rate linearly
(:266-272) and runs the fused AdamW + EMA step (optim.FusedAdamWEMA, one launch instead of :246-256).

Logging, checkpoint cadence and the data iterator stay with the caller (out of scope); the quantities
the reference logs (`loss`, `mse`, `nll` weighted means, `grad_norm`) are returned as device tensors so that a caller
may log them without forcing a host sync per element.
"""
import contextlib

import torch

from .models.step_sample import LossAwareSampler, UniformSampler
from .ops import to_device_async
from .optim import FusedAdamWEMA


class TrainStep:
    """forward_backward + optimize of utils/train_util.py for one model replica.

    :param model: TransformerNetModel on the GPU (train mode is the caller's choice, as in the reference).
    :param diffusion: GaussianDiffusion / SpacedDiffusion.
    :param microbatch: micro-batch size; <= 0 means the whole batch (train_util.py:68).
    :param schedule_sampler: a step_sample sampler (default UniformSampler, :84).
    :param ddp_model: the DistributedDataParallel shell around `model` when a process group is up (:106-116);
                      None = single replica (`use_ddp = False`, :118-119).
    """

    def __init__(self, model, diffusion, microbatch=-1, lr=1e-4, weight_decay=0.0, ema_rate=(0.9999,), learning_steps=0,
                 gradient_clipping=-1.0, schedule_sampler=None, ddp_model=None, resume_step=0, optimizer=None):
        self.model, self.diffusion = model, diffusion
        self.ddp_model = ddp_model if ddp_model is not None else model
        self.use_ddp = ddp_model is not None
        self.microbatch = microbatch
        self.lr, self.learning_steps, self.gradient_clipping = float(lr), int(learning_steps), float(gradient_clipping)
        self.schedule_sampler = schedule_sampler or UniformSampler(diffusion)
        # every rank is built with the same `microbatch`: the loss-aware sampler sizes its gather block from it, so ranks whose last
        # micro-batch is shorter (or whose `data` iterators yield different batch sizes) still enter the collective with equal buffers
        if isinstance(self.schedule_sampler, LossAwareSampler) and microbatch > 0 and self.schedule_sampler.max_local_batch is None:
            self.schedule_sampler.max_local_batch = int(microbatch)
        # train_util.py:63-67: a float, or the config's comma-separated string ("0.5,0.9,0.99"); a sequence of floats is accepted
        # too, None (the reference's empty list) means no EMA copies
        # the reference's falsy rule comes first (`... if ema_rate else []`): 0.0, "", None and () all mean no EMA copies
        if not ema_rate:
            self.ema_rate = []
        elif isinstance(ema_rate, (int, float)):
            self.ema_rate = [float(ema_rate)]
        elif isinstance(ema_rate, str):
            self.ema_rate = [float(x) for x in ema_rate.split(",") if x.strip()]
        else:
            self.ema_rate = [float(r) for r in ema_rate]
        self.model_params = list(model.parameters())
        # `optimizer`: anything with grad_norm() / step(lr=) (tests inject a host stand-in; the product is the fused kernel)
        self.opt = optimizer or FusedAdamWEMA(self.model_params, lr=self.lr, weight_decay=weight_decay, ema_rates=self.ema_rate)
        self.step, self.resume_step = 0, int(resume_step)
        self.last_losses = {}

    # ------------------------------------------------------------------ train_util.py:188-238
    def zero_grad(self):
        """train_util.py:240-244 zero-fills every gradient in place; here the tensors are dropped instead, so the first micro-batch's
        backward stores its gradients rather than adding them to zeros (same values; 187 fills + 187 adds less per step at the
        reference's model size).  The optimizer's device table follows the new addresses (optim.FusedAdamWEMA._tensor_table)."""
        for p in self.model_params:
            p.grad = None

    def _forward_backward_logic(self, cond, backward):
        self.zero_grad()
        prev_train_mode, prev_grad_mode = self.model.training, torch.is_grad_enabled()
        if not backward:
            self.model.eval()
            torch.set_grad_enabled(False)
        dev = self.model_params[0].device
        n = cond["input_ids"].shape[0]
        micro = self.microbatch if self.microbatch > 0 else n
        sums, count = {}, 0
        try:
            for i in range(0, n, micro):
                micro_cond = {k: to_device_async(v[i:i + micro], dev) for k, v in cond.items()}
                last_batch = (i + micro) >= n
                t, weights = self.schedule_sampler.sample(micro_cond["input_ids"].shape[0], dev)
                sync = contextlib.nullcontext() if (last_batch or not self.use_ddp) else self.ddp_model.no_sync()
                with sync:
                    losses = self.diffusion.training_losses(self.ddp_model, t, model_kwargs=micro_cond)
                    if backward:
                        if isinstance(self.schedule_sampler, LossAwareSampler):
                            self.schedule_sampler.update_with_local_losses(t, losses["loss"].detach())
                        (losses["loss"] * weights).mean().backward()
                for k, v in losses.items():
                    sums[k] = sums.get(k, 0) + (v.detach() * weights).mean()
                count += 1
        finally:
            if not backward:
                self.model.train(prev_train_mode)
                torch.set_grad_enabled(prev_grad_mode)
        self.last_losses = {("" if backward else "eval_") + k: v / count for k, v in sums.items()}
        return self.last_losses

    def forward_only(self, cond):
        return self._forward_backward_logic(cond, backward=False)

    def forward_backward(self, cond):
        return self._forward_backward_logic(cond, backward=True)

    # ------------------------------------------------------------------ train_util.py:246-272
    def _anneal_lr(self):
        if not self.learning_steps:
            return self.lr
        frac_done = (self.step + self.resume_step) / self.learning_steps
        return self.lr * (1 - frac_done)

    def optimize(self):
        if self.gradient_clipping > 0:            # train_util.py:248-249, :255-264 (`grad_clip`)
            if hasattr(self.opt, "clip_grad_norm"):   # the fused optimizer: the library's norm + one scale kernel, no host sync
                self.opt.clip_grad_norm(self.gradient_clipping)
            else:                                     # an injected optimizer without one: the reference's fallback
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.gradient_clipping)
        # device scalar (the reference logs it, :274-280); a copy: the optimizer's norm buffer is overwritten by the next step
        grad_norm = self.opt.grad_norm().clone()
        self.opt.step(lr=self._anneal_lr())
        return grad_norm

    def run_step(self, cond):
        """One iteration of run_loop's body (:170-172, :185): forward_backward, optimize, step += 1."""
        losses = self.forward_backward(cond)
        grad_norm = self.optimize()
        self.step += 1
        return losses, grad_norm
