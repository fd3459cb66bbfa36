"""Training-time timestep samplers behind the names `run/train.py:22` and `utils/train_util.py:17` import
(`create_named_schedule_sampler`, `UniformSampler`, `LossAwareSampler`, ...; MuseDiffusion/models/step_sample.py).

Everything here is host arithmetic over T (= a few thousand) float64 numbers, as in the reference, so that a seeded
`np.random` run picks the same timesteps.  What differs is the loss-aware sampler's exchange between ranks: the
reference pads and all-gathers batch sizes, timesteps and losses separately and then reads every element with
`.item()` (step_sample.py:100-122); here a rank ships its (timesteps, losses) as one [2, max_local_batch] float64
block in a single all_gather and the host unpacks it once - later, when the sampler's state is next read, so that the copy to
the host runs under the backward pass instead of stalling the launch queue between forward and backward.
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
    from ..ops import to_device_async      # (a pinned staging copy: the draw must not drain the GPU's queue once per step)
    return (to_device_async(torch.as_tensor(t, dtype=torch.long), device), to_device_async(torch.as_tensor(importance, dtype=torch.float), device))


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
    (timestep, loss) pairs in rank order and then applies `update_with_all_losses` - identical state everywhere.

    No host round trip between the forward and the backward (the reference's version, step_sample.py:90-122, runs two
    all_gathers, `.item()` per element and Python loops right there, every micro-batch): with device tensors the call only ENQUEUES
    - one packed all_gather (world > 1) and one device -> pinned-host copy of [count, timesteps, losses] - and returns; the host
    applies `update_with_all_losses` when the state is next read (`weights()` / `sample()` of the next micro-batch, or the
    `_loss_history` / `_loss_counts` attributes), by which time the copy has finished under the backward pass.  The order of
    updates and every value are the reference's: nothing reads the state in between."""

    _pending = None
    _GROUP = 64       # gather blocks are padded to a multiple of this many (timestep, loss) pairs
    # A bound on the micro-batch size that EVERY rank agrees on (train_step.TrainStep sets its configured `microbatch`).  The gather
    # block of the device path is sized from it, so all ranks enter the collective with equal buffers whatever their own counts are.
    # None: the ranks exchange their counts first and pad to the largest, as the reference does (step_sample.py:100-113; one host
    # read per micro-batch).
    max_local_batch = None

    def update_with_all_losses(self, ts, losses):
        raise NotImplementedError

    def _flush(self):
        """apply the update whose copy was enqueued by the last update_with_local_losses (no-op when there is none)"""
        pend, self._pending = self._pending, None
        if pend is None:
            return
        host, event, world, cap = pend
        event.synchronize()
        rows = host.view(world, 1 + 2 * cap).numpy()
        all_ts, all_losses = [], []
        for r in range(world):
            n = int(rows[r, 0])
            if n > cap:   # a rank's micro-batch exceeded the bound every rank pads to: all ranks see it here and fail together
                raise ValueError("loss-aware sampler: rank %d brought a micro-batch of %d rows, more than max_local_batch allows "
                                 "(padded size %d)" % (r, n, cap))
            all_ts.extend(int(v) for v in rows[r, 1:1 + n])
            all_losses.extend(float(v) for v in rows[r, 1 + cap:1 + cap + n])
        self.update_with_all_losses(all_ts, all_losses)

    def update_with_local_losses(self, local_ts, local_losses):
        self._flush()                                   # updates apply in call order
        ts64 = local_ts.detach().to(torch.float64)
        ls64 = local_losses.detach().to(device=local_ts.device, dtype=torch.float64)
        world = dist.get_world_size() if dist.is_initialized() else 1
        count = int(local_ts.numel())
        if not local_ts.is_cuda:                         # host tensors (CPU tests, gloo): nothing to overlap with
            if world == 1:
                self.update_with_all_losses([int(v) for v in ts64.tolist()], ls64.tolist())
                return
            counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(counts, torch.tensor([count], dtype=torch.int64))
            counts = [int(c) for c in counts]
            block = torch.zeros(2, max(counts), dtype=torch.float64)
            block[0, :count], block[1, :count] = ts64, ls64
            blocks = [torch.zeros_like(block) for _ in range(world)]
            dist.all_gather(blocks, block)
            all_ts, all_losses = [], []
            for blk, n in zip(blocks, counts):
                all_ts.extend(int(v) for v in blk[0, :n].tolist())
                all_losses.extend(blk[1, :n].tolist())
            self.update_with_all_losses(all_ts, all_losses)
            return
        dev = local_ts.device
        bound = self.max_local_batch if world > 1 else count
        if bound is None:      # no agreed bound: learn the largest count first (the reference's size exchange)
            sizes = torch.empty(world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(sizes, torch.full((1,), count, dtype=torch.int64, device=dev))
            bound = int(sizes.max())
        cap = (max(int(bound), 1) + self._GROUP - 1) // self._GROUP * self._GROUP
        if count > cap and world == 1:
            raise ValueError("loss-aware sampler: a micro-batch of %d rows exceeds max_local_batch = %d, the size every rank "
                             "pads its gather block to" % (count, bound))
        # world > 1 and this rank over the bound: the failure must be COLLECTIVE - a rank that raised here would leave the others
        # waiting in the all_gather until the RCCL timeout.  It enters the collective with its true count in the header (and the pairs
        # that fit); every rank then finds that count above the padded size when it unpacks (_flush) and raises there.
        fit = min(count, cap)
        block = torch.zeros(1 + 2 * cap, dtype=torch.float64, device=dev)
        block[0:1].fill_(float(count))      # (a fill kernel: `block[0] = count` would stage the scalar through a pageable host tensor and drain the stream)
        block[1:1 + fit] = ts64[:fit]
        block[1 + cap:1 + cap + fit] = ls64[:fit]
        if world > 1:
            gathered = torch.empty(world * (1 + 2 * cap), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(gathered, block)     # one collective per micro-batch; stream-ordered under RCCL
        else:
            gathered = block
        host = torch.empty(gathered.shape, dtype=torch.float64).pin_memory()
        host.copy_(gathered, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        self._keep = gathered                                # (alive until the copy has run)
        self._pending = (host, event, world, cap)


class LossSecondMomentResampler(LossAwareSampler):
    """p_t proportional to sqrt(mean of the last `history_per_term` squared losses seen at t), blended with a uniform
    floor of `uniform_prob`; uniform until every timestep has a full window (step_sample.py:143-173).
    `_loss_history[t]` keeps the window oldest-first and `_loss_counts[t]` its fill, under the reference's names."""

    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        T = diffusion.num_timesteps
        self._history = np.zeros([T, history_per_term], dtype=np.float64)
        self._counts = np.zeros([T], dtype=int)

    # the reference's attribute names; reading them first applies an update that is still on its way from the device
    @property
    def _loss_history(self):
        self._flush()
        return self._history

    @_loss_history.setter
    def _loss_history(self, value):
        self._flush()
        self._history = value

    @property
    def _loss_counts(self):
        self._flush()
        return self._counts

    @_loss_counts.setter
    def _loss_counts(self, value):
        self._flush()
        self._counts = value

    def _warmed_up(self):
        """step_sample.py:172-173"""
        return bool((self._loss_counts == self.history_per_term).all())

    def weights(self):
        T = self.diffusion.num_timesteps
        if not self._warmed_up():
            return np.ones([T], dtype=np.float64)
        rms = np.sqrt(np.mean(self._loss_history ** 2, axis=-1))
        rms /= np.sum(rms)
        rms *= 1 - self.uniform_prob
        rms += self.uniform_prob / len(rms)
        return rms

    def update_with_all_losses(self, ts, losses):
        window, counts = self._history, self._counts
        for t, value in zip(ts, losses):
            filled = counts[t]
            if filled < self.history_per_term:
                window[t, filled] = value
                counts[t] = filled + 1
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
