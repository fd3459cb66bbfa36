"""Fused AdamW + EMA optimizer step.

The reference's `TrainLoop.optimize` (utils/train_util.py:246-280) runs torch.optim.AdamW over ~200 parameter
tensors, then `update_ema` (:21-31) over each of 3 EMA copies, then a grad-norm loop with one `.item()` per
tensor.  Here one kernel launch (`mh_adamw_ema_step`) updates every parameter, its Adam moments and all EMA
copies, and one more (`mh_grad_norm`) produces the gradient norm on the device.

`state_dict()` / `load_state_dict()` use torch.optim.AdamW's format, and `ema_state_dict(i, model)` returns an
EMA copy under the model's state_dict keys, so checkpoints stay in the reference's on-disk format
(`model_*.pt`, `ema_{rate}_*.pt`, `opt_*.pt`; utils/train_util.py:294-319).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import OptHParams, check, current_stream, lib, ptr

CHUNK = 1 << 16


class FusedAdamWEMA:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_rates=()):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        if len(ema_rates) > 4:
            raise ValueError("at most 4 EMA rates")
        for p in self.params:
            _lib.require_device(p)
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("parameters must be contiguous fp32 tensors")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.ema_rates = [float(r) for r in ema_rates]
        self.step_count = 0
        dev = self.params[0].device
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.ema = [[p.detach().clone() for p in self.params] for _ in self.ema_rates]   # copy.deepcopy(master_params), :102
        chunks = []
        for i, p in enumerate(self.params):
            n = p.numel()
            for off in range(0, n, CHUNK):
                chunks.append((i, min(CHUNK, n - off), off))
        arr = np.zeros(len(chunks), dtype=[("tensor", "<i4"), ("count", "<i4"), ("offset", "<i8")])
        for k, (i, c, o) in enumerate(chunks):
            arr[k] = (i, c, o)
        self.n_chunks = len(chunks)
        self._chunks = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
        self._partial = torch.empty(self.n_chunks, device=dev, dtype=torch.float32)
        self._norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self._table = None
        self._table_key = None
        self._stage, self._stage_ev, self._stage_k = None, None, 0
        # torch keeps optimizer state only for parameters that have taken a step (a frozen one never appears in `state`)
        self._stepped = [False] * len(self.params)
        self._steps = [0] * len(self.params)     # per-parameter step numbers, as torch.optim.AdamW keeps them (state[p]["step"])
        self._with_grad = [True] * len(self.params)

    def _tensor_table(self):
        # torch.optim.AdamW skips a parameter whose gradient is None (frozen by requires_grad_(False) - the reference's
        # --freeze_embedding, utils/initialization.py:64-65 - or never reached by the backward); update_ema still walks it
        # (train_util.py:21-31, :253-254).  Such a row carries a NULL gradient pointer and the kernels do the same.
        grads = [p.grad if p.requires_grad else None for p in self.params]
        if all(g is None for g in grads):
            raise RuntimeError("no parameter has a gradient: call backward() before the optimizer step")
        for i, g in enumerate(grads):
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError("gradients must be contiguous fp32")
            if g.data_ptr() & 15:     # the kernels read 16-byte pieces: an autograd view at an odd offset is copied once
                self.params[i].grad = grads[i] = g.clone()
        key = tuple(0 if g is None else g.data_ptr() for g in grads)
        if self._table is None or key != self._table_key:
            rows = np.zeros((len(self.params), 8), dtype=np.int64)
            for i, p in enumerate(self.params):
                g = grads[i]
                rows[i, 0], rows[i, 1] = p.data_ptr(), 0 if g is None else g.data_ptr()
                rows[i, 2], rows[i, 3] = self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr()
                for e in range(len(self.ema_rates)):
                    rows[i, 4 + e] = self.ema[e][i].data_ptr()
            if (rows[:, [0, 2, 3] + [4 + e for e in range(len(self.ema_rates))]] & 15).any():
                raise ValueError("parameters, moments and EMA copies must be 16-byte aligned")
            # gradients that were dropped and re-created move: the table follows through a pinned staging buffer and a
            # stream-ordered copy (no host synchronisation); two staging buffers, each re-used only once its last copy has run
            if self._stage is None:
                self._stage = [torch.empty(rows.shape, dtype=torch.int64).pin_memory() for _ in range(2)]
                self._stage_ev = [None, None]
                self._table = torch.empty(rows.shape, dtype=torch.int64, device=self.params[0].device)
            k = self._stage_k = 1 - self._stage_k
            if self._stage_ev[k] is not None:
                self._stage_ev[k].synchronize()
            self._stage[k].copy_(torch.from_numpy(rows))
            self._table.copy_(self._stage[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._stage_ev[k] = ev
            self._table_key = key
        self._with_grad = [g is not None for g in grads]
        return self._table

    def grad_norm(self):
        """sqrt(sum g^2) as a 1-element device tensor (train_util.py:274-280 without the per-tensor .item())."""
        t = self._tensor_table()
        check(lib().mh_grad_norm(ptr(t), ptr(self._chunks), self.n_chunks, ptr(self._partial), ptr(self._norm),
                                 current_stream()), "mh_grad_norm")
        return self._norm

    def clip_grad_norm(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ (the reference's `grad_clip`, train_util.py:255-264) without a host sync: the norm kernel,
        then one launch that scales every gradient by min(1, max_norm / (norm + 1e-6)).  Returns the pre-clip norm (device scalar)."""
        norm = self.grad_norm()
        check(lib().mh_clip_grads(ptr(self._tensor_table()), ptr(self._chunks), self.n_chunks, ptr(norm), float(max_norm), current_stream()),
              "mh_clip_grads")
        return norm

    @torch.no_grad()
    def step(self, lr=None):
        """One optimizer step: AdamW on every parameter + all EMA copies, one launch."""
        if lr is not None:
            self.lr = float(lr)
        t = self._tensor_table()
        # one launch applies ONE bias correction: every parameter that steps now must have taken every earlier step too.  A parameter
        # that gets its first gradient later (unfrozen mid-run) would need its own correction, as torch.optim.AdamW's per-parameter
        # `step` gives it - refused rather than stepped with the wrong one (the reference's TrainLoop never changes what is frozen)
        behind = [i for i, w in enumerate(self._with_grad) if w and self._steps[i] != self.step_count]
        if behind:
            raise NotImplementedError("FusedAdamWEMA: parameter %d has taken %d of the optimizer's %d steps - parameters that start to "
                                      "receive gradients mid-run need per-parameter bias correction" % (behind[0], self._steps[behind[0]], self.step_count))
        self.step_count += 1
        for i, w in enumerate(self._with_grad):
            self._stepped[i] = self._stepped[i] or w
            if w:
                self._steps[i] = self.step_count
        b1, b2 = self.betas
        hp = OptHParams()
        hp.beta1, hp.beta2, hp.eps = b1, b2, self.eps
        hp.one_minus_beta1, hp.one_minus_beta2 = 1.0 - b1, 1.0 - b2
        hp.decay_mul = 1.0 - self.lr * self.weight_decay
        hp.step_size = self.lr / (1.0 - b1 ** self.step_count)
        hp.bias2_sqrt = math.sqrt(1.0 - b2 ** self.step_count)
        hp.n_ema = len(self.ema_rates)
        for e, r in enumerate(self.ema_rates):
            hp.ema_rate[e] = r
            hp.ema_one_minus[e] = 1.0 - r
        check(lib().mh_adamw_ema_step(ptr(t), ptr(self._chunks), self.n_chunks, C.byref(hp), current_stream()),
              "mh_adamw_ema_step")
        # the kernel wrote the parameters behind autograd's back: bump their version counters (no launch) so the
        # packed weight arena (network.TransformerNetModel.engine) is rebuilt on the next inference call
        moved = tuple(p for p, w in zip(self.params, self._with_grad) if w)
        torch._C._autograd._unsafe_set_version_counter(moved, tuple(p._version + 1 for p in moved))

    def zero_grad(self, set_to_none=False):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    # ------------------------------------------------------------------ checkpoint format of the reference
    def state_dict(self):
        """torch.optim.AdamW layout (what `opt_{step}.pt` holds, train_util.py:310-315)."""
        state = {i: {"step": torch.tensor(float(self._steps[i])), "exp_avg": self.exp_avg[i], "exp_avg_sq": self.exp_avg_sq[i]}
                 for i in range(len(self.params)) if self._stepped[i]}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])
        for i, st in sd["state"].items():
            i = int(i)
            self.exp_avg[i].copy_(st["exp_avg"])
            self.exp_avg_sq[i].copy_(st["exp_avg_sq"])
            self._steps[i] = int(float(st["step"]))
            self.step_count = max(self.step_count, self._steps[i])
            self._stepped[i] = True

    def ema_state_dict(self, index, model):
        """EMA copy `index` under the model's state_dict keys (`_master_params_to_state_dict`, train_util.py:321-333)."""
        sd = model.state_dict()
        by_ptr = {p.data_ptr(): i for i, p in enumerate(self.params)}
        for name, p in model.named_parameters():
            sd[name] = self.ema[index][by_ptr[p.data_ptr()]]
        # `lm_head.weight` is not a key of named_parameters() while it is tied to the embedding (one Parameter, listed once): the
        # reference's _master_params_to_state_dict (train_util.py:321-333) then leaves the MODEL's tensor under that key, which is the
        # live embedding - kept as it is.  After overload_embedding (utils/initialization.py:61-63) the two are different Parameters,
        # both named, and each EMA copy goes under its own key.
        return sd
