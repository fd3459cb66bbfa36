"""The four training corruptions of MuseDiffusion/data/corruption.py:100-195 as HIP kernels over a whole ragged batch.

A batch is (`values`, `offsets`): every sequence back to back in one int32 device tensor plus int64 row starts
([B + 1]).  Each function takes its random draws as optional arguments; given the draws the reference made (in
the order it made them) the result is bit-identical to the reference's (tests/test_batch_gpu.py against
tests/golden/batch.npz).  Without them the draws come from torch's device generator - same distribution, and no
per-token Python loop (the reference spends O(tokens) interpreter steps per sequence, corruption.py:108-114, :128-133).
"""
import random as _random

import torch

from .._lib import check, current_stream, lib, ptr, require_device

generator = _random.Random()   # drives the host-side choices of Corruptions.__call__ (which corruptions, in which order)


def _prep(values, offsets):
    require_device(values, offsets)
    values = values.to(torch.int32).contiguous()
    offsets = offsets.to(torch.int64).contiguous()
    B = offsets.numel() - 1
    return values, offsets, B


def _check_rows(offsets):
    longest = int((offsets[1:] - offsets[:-1]).max()) if offsets.numel() > 1 else 0
    if longest > int(lib().mh_batch_max_row()):
        raise ValueError("sequence of %d tokens exceeds the %d-token row limit of the batch kernels" % (longest, lib().mh_batch_max_row()))


def masking_token(values, offsets, p=0.3, u=None):
    """corruption.py:100-114.  u[offsets[b] + k]: the draw for position 12 + k of row b."""
    values, offsets, B = _prep(values, offsets)
    u = torch.rand(values.numel(), device=values.device) if u is None else u.to(values.device, torch.float32).contiguous()
    out = torch.empty_like(values)
    check(lib().mh_corrupt_masking_token(ptr(values), ptr(offsets), ptr(u), float(p), ptr(out), B, current_stream()), "mh_corrupt_masking_token")
    return out


def masking_note(values, offsets, p=0.5, u=None):
    """corruption.py:117-133.  u[offsets[b] + k]: the draw for the k-th velocity token of row b that has three tokens after it."""
    values, offsets, B = _prep(values, offsets)
    _check_rows(offsets)
    u = torch.rand(values.numel(), device=values.device) if u is None else u.to(values.device, torch.float32).contiguous()
    out = torch.empty_like(values)
    check(lib().mh_corrupt_masking_note(ptr(values), ptr(offsets), ptr(u), float(p), ptr(out), B, current_stream()), "mh_corrupt_masking_note")
    return out


def randomize_note(values, offsets, p=0.5, u=None, new_tokens=None):
    """corruption.py:136-162.  new_tokens[offsets[b] + k] = (velocity 131..194, pitch 3..130, duration 304..431) for draw k."""
    values, offsets, B = _prep(values, offsets)
    _check_rows(offsets)
    n, dev = values.numel(), values.device
    u = torch.rand(n, device=dev) if u is None else u.to(dev, torch.float32).contiguous()
    if new_tokens is None:
        new_tokens = torch.stack([torch.randint(131, 195, (n,), device=dev), torch.randint(3, 131, (n,), device=dev),
                                  torch.randint(304, 432, (n,), device=dev)], dim=1)
    new_tokens = new_tokens.to(dev, torch.int32).contiguous()
    out = torch.empty_like(values)
    check(lib().mh_corrupt_randomize_note(ptr(values), ptr(offsets), ptr(u), ptr(new_tokens), float(p), ptr(out), B, current_stream()),
          "mh_corrupt_randomize_note")
    return out


def _draw_bar_pairs(values, offsets, count):
    """two distinct bar indices per swap, sorted - random.sample(range(n_bars), 2) of corruption.py:177, drawn on the device"""
    B = offsets.numel() - 1
    rows = torch.repeat_interleave(torch.arange(B, device=values.device), offsets[1:] - offsets[:-1])
    nbar = torch.bincount(rows[values == 2], minlength=B).clamp(min=2).unsqueeze(1)
    r = torch.rand(B, count, 2, device=values.device)
    first = (r[..., 0] * nbar).long().clamp(max=nbar - 1)
    second = (r[..., 1] * (nbar - 1)).long().clamp(max=nbar - 2)
    second = second + (second >= first).long()
    return torch.stack([torch.minimum(first, second), torch.maximum(first, second)], dim=-1).to(torch.int32)


def random_rotating(values, offsets, count=3, pairs=None, return_status=False):
    """corruption.py:165-195.  pairs [B, count, 2]: the sorted bar indices of each swap."""
    values, offsets, B = _prep(values, offsets)
    _check_rows(offsets)
    pairs = _draw_bar_pairs(values, offsets, count) if pairs is None else pairs
    pairs = pairs.to(values.device, torch.int32).contiguous()
    assert pairs.shape == (B, count, 2)
    out = torch.empty_like(values)
    status = torch.empty(B, device=values.device, dtype=torch.int32)
    check(lib().mh_corrupt_random_rotating(ptr(values), ptr(offsets), ptr(pairs), int(count), ptr(out), ptr(status), B, current_stream()),
          "mh_corrupt_random_rotating")
    return (out, status) if return_status else out


class Corruptions:
    """Same configuration surface as the reference's class (corruption.py:11-57: `corr_available`, `corr_max`, `corr_p`,
    `corr_kwargs`); `__call__(values, offsets)` corrupts a whole ragged batch on the device.  As in the reference the
    available corruptions are shuffled, the first `corr_max` are kept and each is applied with probability `corr_p` -
    decided once per call (per batch), where the reference decides per sequence."""

    MAP = {"mt": (masking_token, {"p": 0.3}), "mn": (masking_note, {"p": 0.5}), "rn": (randomize_note, {"p": 0.5}),
           "rr": (random_rotating, {"count": 3})}

    @classmethod
    def from_config(cls, corr_available, corr_max, corr_p, corr_kwargs=None):
        return cls(tuple(corr_available.split(",")), int(corr_max), float(corr_p), eval(corr_kwargs) if corr_kwargs else None)

    def __init__(self, corr_available, corr_max, corr_p, corr_kwargs=None):
        assert all(k in self.MAP for k in corr_available), corr_available
        assert 0 <= corr_max <= len(corr_available) and 0 <= corr_p <= 1
        assert corr_kwargs is None or isinstance(corr_kwargs, dict)
        self.corr_available, self.corr_max, self.corr_p, self.corr_kwargs = tuple(corr_available), corr_max, corr_p, corr_kwargs

    def __call__(self, values, offsets):
        order = list(self.corr_available)
        generator.shuffle(order)
        out = values
        for key in order[: self.corr_max]:
            if generator.random() > 1 - self.corr_p:
                fn, defaults = self.MAP[key]
                kw = {k: (self.corr_kwargs or {}).get(k, v) for k, v in defaults.items()}
                out = fn(out, offsets, **kw)
        return out

    def __repr__(self):
        return "Corruptions(corr_available=[%s], corr_max=%r, corr_p=%r, corr_kwargs=%r)" % (
            ",".join(self.corr_available), self.corr_max, self.corr_p, self.corr_kwargs)
