"""Training-time timestep samplers behind the names `run/train.py:22` and `utils/train_util.py:17` import
(`create_named_schedule_sampler`, `UniformSampler`, `LossAwareSampler`, ...; MuseDiffusion/models/step_sample.py).

Everything here is host arithmetic over T (= a few thousand) float64 numbers, as in the reference, so that a seeded
`np.random` run picks the same timesteps.  What differs is the loss-aware sampler's exchange between ranks: the
reference pads and all-gathers batch sizes, timesteps and losses separately and then reads every element with
`.item()` (step_sample.py:100-122); here a rank ships its (timesteps, losses) as one [2, max_local_batch] float64
block in a single all_gather (plus the usual size exchange) and unpacks on the host once.
"""
import numpy as np
import torch
import torch.distributed as dist


def _draw_timesteps(weights, batch_size, device):
    """Importance-sample `batch_size` timesteps from unnormalised `weights`.  Returns (t int64 [B], 1 / (T p_t) fp32 [B])
    - the same single `np.random.choice` call as step_sample.py:45-63, so host seeds reproduce the reference's picks."""
    prob = weights / np.sum(weights)
    t = np.random.choice(len(prob), size=(batch_size,), p=prob)
    importance = 1 / (len(prob) * prob[t])
    return (torch.as_tensor(t, device=device, dtype=torch.long), torch.as_tensor(importance, device=device, dtype=torch.float))


class ScheduleSampler:
    """Base: anything with `weights()` -> np.ndarray [T] of positive numbers can `sample(batch_size, device)`."""

    def weights(self):
        raise NotImplementedError

    def sample(self, batch_size, device):
        return _draw_timesteps(self.weights(), batch_size, device)


class _TableSampler(ScheduleSampler):
    """A sampler whose weights never change."""

    def __init__(self, diffusion, table):
        self.diffusion = diffusion
        self._weights = np.asarray(table, dtype=np.float64)

    def weights(self):
        return self._weights


class UniformSampler(_TableSampler):
    """Every timestep equally likely (step_sample.py:68-74)."""

    def __init__(self, diffusion):
        super().__init__(diffusion, np.ones([diffusion.num_timesteps]))


class FixSampler(_TableSampler):
    """Timesteps of the first half twice as likely as those of the second (step_sample.py:76-87)."""

    def __init__(self, diffusion):
        n = diffusion.num_timesteps // 2
        super().__init__(diffusion, np.concatenate([np.full([n], 1.0), np.full([n], 0.5)]))


class LossAwareSampler(ScheduleSampler):
    """Samplers that adapt to the training losses.  `update_with_local_losses` makes every rank see every rank's
    (timestep, loss) pairs in rank order and then applies `update_with_all_losses` - identical state everywhere."""

    def update_with_all_losses(self, ts, losses):
        raise NotImplementedError

    def update_with_local_losses(self, local_ts, local_losses):
        ts64 = local_ts.detach().to(torch.float64)
        ls64 = local_losses.detach().to(torch.float64)
        if not dist.is_initialized() or dist.get_world_size() == 1:
            self.update_with_all_losses([int(v) for v in ts64.tolist()], ls64.tolist())
            return
        world, dev, count = dist.get_world_size(), local_ts.device, int(local_ts.numel())
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([count], dtype=torch.int64, device=dev))
        counts = [int(c) for c in counts]
        block = torch.zeros(2, max(counts), dtype=torch.float64, device=dev)
        block[0, :count], block[1, :count] = ts64, ls64
        blocks = [torch.zeros_like(block) for _ in range(world)]
        dist.all_gather(blocks, block)
        all_ts, all_losses = [], []
        for blk, n in zip(blocks, counts):
            host = blk[:, :n].cpu()
            all_ts.extend(int(v) for v in host[0].tolist())
            all_losses.extend(host[1].tolist())
        self.update_with_all_losses(all_ts, all_losses)


class LossSecondMomentResampler(LossAwareSampler):
    """p_t proportional to sqrt(mean of the last `history_per_term` squared losses seen at t), blended with a uniform
    floor of `uniform_prob`; uniform until every timestep has a full window (step_sample.py:143-173).
    `_loss_history[t]` keeps the window oldest-first and `_loss_counts[t]` its fill, under the reference's names."""

    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        T = diffusion.num_timesteps
        self._loss_history = np.zeros([T, history_per_term], dtype=np.float64)
        self._loss_counts = np.zeros([T], dtype=int)

    def weights(self):
        T = self.diffusion.num_timesteps
        if (self._loss_counts < self.history_per_term).any():
            return np.ones([T], dtype=np.float64)
        rms = np.sqrt(np.mean(self._loss_history ** 2, axis=-1))
        rms /= np.sum(rms)
        rms *= 1 - self.uniform_prob
        rms += self.uniform_prob / len(rms)
        return rms

    def update_with_all_losses(self, ts, losses):
        window = self._loss_history
        for t, value in zip(ts, losses):
            filled = self._loss_counts[t]
            if filled < self.history_per_term:
                window[t, filled] = value
                self._loss_counts[t] = filled + 1
            else:                                   # slide: forget the oldest entry
                window[t] = np.append(window[t, 1:], value)


_SAMPLERS = {"uniform": UniformSampler, "fixstep": FixSampler, "lossaware": LossSecondMomentResampler}


def create_named_schedule_sampler(name, diffusion):
    """Factory of step_sample.py:11-27: 'uniform' | 'fixstep' | 'lossaware' (the latter needs a process group)."""
    if name not in _SAMPLERS:
        raise NotImplementedError(f"unknown schedule sampler: {name}")
    if name == "lossaware" and not dist.is_initialized():
        raise RuntimeError("Cannot use lossaware sampler without distributed runtime.")
    return _SAMPLERS[name](diffusion)
