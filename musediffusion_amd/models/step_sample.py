"""Timestep samplers for training, API of MuseDiffusion/models/step_sample.py.

Host-side numpy like the reference (a few thousand floats); the only change is the loss-aware
sampler's synchronisation: the reference issues three all_gathers and one `.item()` per element
per micro-batch (step_sample.py:100-122); here each rank contributes ONE packed [2, max_bs + 1]
fp64 buffer (count, timesteps, losses) to a single all_gather over RCCL / gloo.
"""
from abc import ABC, abstractmethod

import numpy as np
import torch
import torch.distributed as dist


def create_named_schedule_sampler(name, diffusion):
    """step_sample.py:11-27."""
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "fixstep":
        return FixSampler(diffusion)
    if name == "lossaware":
        if not dist.is_initialized():
            raise RuntimeError("Cannot use lossaware sampler without distributed runtime.")
        return LossSecondMomentResampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")


class ScheduleSampler(ABC):
    """A distribution over timesteps used for importance sampling (step_sample.py:30-65)."""

    @abstractmethod
    def weights(self):
        """Positive (not necessarily normalised) weight per diffusion step."""

    def sample(self, batch_size, device):
        """(timesteps int64 [B], weights fp32 [B]) with weights 1 / (T p_t)."""
        w = self.weights()
        p = w / np.sum(w)
        picked = np.random.choice(len(p), size=(batch_size,), p=p)
        indices = torch.as_tensor(picked, device=device, dtype=torch.long)
        weights = torch.as_tensor(1 / (len(p) * p[picked]), device=device, dtype=torch.float)
        return indices, weights


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class FixSampler(ScheduleSampler):
    """First half weight 1, second half 0.5 (step_sample.py:76-87)."""

    def __init__(self, diffusion):
        self.diffusion = diffusion
        half = diffusion.num_timesteps // 2
        self._weights = np.concatenate([np.ones([half]), np.zeros([half]) + 0.5])

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    def update_with_local_losses(self, local_ts, local_losses):
        """Gather every rank's (timesteps, losses) and update identically everywhere
        (step_sample.py:90-123), with one collective."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        n = int(local_ts.numel())
        if world == 1:
            self.update_with_all_losses(local_ts.detach().cpu().tolist(), local_losses.detach().cpu().tolist())
            return
        dev = local_ts.device
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=dev))
        max_bs = max(int(s) for s in sizes)
        pack = torch.zeros(2, max_bs, dtype=torch.float64, device=dev)
        pack[0, :n] = local_ts.to(torch.float64)
        pack[1, :n] = local_losses.detach().to(torch.float64)
        gathered = [torch.zeros_like(pack) for _ in range(world)]
        dist.all_gather(gathered, pack)
        ts, losses = [], []
        for g, s in zip(gathered, sizes):
            g = g.cpu()
            ts += [int(v) for v in g[0, : int(s)].tolist()]
            losses += g[1, : int(s)].tolist()
        self.update_with_all_losses(ts, losses)

    @abstractmethod
    def update_with_all_losses(self, ts, losses):
        """Deterministic update from the gathered (timestep, loss) pairs."""


class LossSecondMomentResampler(LossAwareSampler):
    """Weights sqrt(E[loss^2]) over the last `history_per_term` losses of each timestep, mixed with
    a small uniform floor, once every timestep has a full history (step_sample.py:143-173)."""

    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        self._loss_history = np.zeros([diffusion.num_timesteps, history_per_term], dtype=np.float64)
        self._loss_counts = np.zeros([diffusion.num_timesteps], dtype=int)

    def weights(self):
        if not self._warmed_up():
            return np.ones([self.diffusion.num_timesteps], dtype=np.float64)
        w = np.sqrt(np.mean(self._loss_history ** 2, axis=-1))
        w /= np.sum(w)
        w *= 1 - self.uniform_prob
        w += self.uniform_prob / len(w)
        return w

    def update_with_all_losses(self, ts, losses):
        for t, loss in zip(ts, losses):
            if self._loss_counts[t] == self.history_per_term:
                self._loss_history[t, :-1] = self._loss_history[t, 1:]   # drop the oldest
                self._loss_history[t, -1] = loss
            else:
                self._loss_history[t, self._loss_counts[t]] = loss
                self._loss_counts[t] += 1

    def _warmed_up(self):
        return (self._loss_counts == self.history_per_term).all()
