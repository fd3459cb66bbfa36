"""Oracle: training_losses (both variants) and the timestep samplers.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch-CPU fp32 with autograd, so
tests can also compare gradients.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .sampling import extract, model_timesteps, q_sample


def mean_flat(x):
    """diffusion.py:15-19."""
    return x.mean(dim=list(range(1, x.dim())))


def token_nll(x, logits_fn, ids, mask=None):
    """diffusion.py:556-575: per-sequence mean (or mask-weighted mean) of token cross-entropy."""
    logits = logits_fn(x)
    nll = F.cross_entropy(logits.view(-1, logits.size(-1)), ids.view(-1).long(),
                          reduction="none").view(ids.shape)
    if mask is not None:
        nll = nll * mask
        return nll.sum(dim=-1) / mask.sum(dim=-1)
    return nll.mean(dim=-1)


def training_losses(d, model_fn, embed_fn, logits_fn, t, input_ids, input_mask,
                    correct_ids=None, noise=None, draws=None):
    """diffusion.py:594-647 (no correct_ids) / :649-699 (with correct_ids).

    RNG order (diffusion.py:614-616 / :665-668): randn(x_start) [, randn(correct_x_start)], randn(noise).
    `draws` (optional dict) injects those tensors under keys 'x_start', 'correct', 'noise'.
    """
    draws = draws or {}
    mean = embed_fn(input_ids)
    std = extract(d.sqrt_one_minus_alphas_cumprod, torch.tensor([0]), mean.shape)

    def jitter(m, key):
        z = draws[key] if key in draws else torch.randn_like(m)
        return m + std * z

    x_start = jitter(mean, "x_start")
    if correct_ids is not None:
        tgt_mean = embed_fn(correct_ids)
        tgt_start = jitter(tgt_mean, "correct")
        tgt_ids = correct_ids
    else:
        tgt_mean, tgt_start, tgt_ids = mean, x_start, input_ids
    if noise is None:
        noise = draws["noise"] if "noise" in draws else torch.randn_like(x_start)
    x_t = q_sample(d, x_start, t, noise=noise, mask=input_mask)
    out = model_fn(x_t, model_timesteps(d, t))
    if d.predict_xstart:
        pred_x0 = out
    else:
        pred_x0 = (extract(d.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                   - extract(d.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * out)
    t_loss = mean_flat((tgt_start - out) ** 2)
    t0_loss = mean_flat((tgt_mean - pred_x0) ** 2)
    mse = torch.where(t == 0, t0_loss, t_loss)
    out_mean = extract(d.sqrt_alphas_cumprod, torch.tensor([d.num_timesteps - 1]), x_start.shape) * x_start
    tT_loss = mean_flat(out_mean ** 2)
    decoder_nll = token_nll(x_start, logits_fn, input_ids)
    nll = token_nll(pred_x0, logits_fn, tgt_ids, mask=input_mask)
    return dict(mse=mse, nll=nll, loss=mse + decoder_nll + tT_loss)


# ---------------------------------------------------------------- timestep samplers


def sample_timesteps(weights, batch_size, rng=np.random):
    """step_sample.py:49-65."""
    p = weights / np.sum(weights)
    idx = rng.choice(len(p), size=(batch_size,), p=p)
    return idx, 1 / (len(p) * p[idx])


class SecondMomentResampler:
    """step_sample.py:143-173."""

    def __init__(self, T, history_per_term=10, uniform_prob=0.001):
        self.T, self.h, self.up = T, history_per_term, uniform_prob
        self.hist = np.zeros([T, history_per_term], dtype=np.float64)
        self.counts = np.zeros([T], dtype=int)

    def weights(self):
        if not (self.counts == self.h).all():
            return np.ones([self.T], dtype=np.float64)
        w = np.sqrt(np.mean(self.hist ** 2, axis=-1))
        w /= np.sum(w)
        w *= 1 - self.up
        w += self.up / len(w)
        return w

    def update(self, ts, losses):
        for t, l in zip(ts, losses):
            if self.counts[t] == self.h:
                self.hist[t, :-1] = self.hist[t, 1:]
                self.hist[t, -1] = l
            else:
                self.hist[t, self.counts[t]] = l
                self.counts[t] += 1
