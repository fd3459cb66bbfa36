#!/usr/bin/env python3
"""Throughput of the device batch producers / token validators at a BASELINE-sized batch
(512 sequences, ~1000 tokens each), with the numpy oracle (= the reference's per-token Python loops restated) beside it."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import data as mdata  # noqa: E402
from musediffusion_amd.utils import decode_util as mdec  # noqa: E402
from oracle import batch as ob  # noqa: E402

rng = np.random.default_rng(0)
rows = []
for _ in range(512):
    seq = list(rng.integers(560, 729, 11)) + [0]
    for k in range(int(rng.integers(200, 252))):
        if k % 24 == 0:
            seq.append(2)
        seq += [int(rng.integers(432, 560)), int(rng.integers(131, 195)), int(rng.integers(3, 131)), int(rng.integers(304, 432))]
    seq.append(1)
    rows.append(np.array(seq, np.int32))
v, off = mdata.to_ragged(rows, "cuda")
n = v.numel()
u = torch.rand(n, device="cuda")
new = torch.stack([torch.randint(131, 195, (n,)), torch.randint(3, 131, (n,)), torch.randint(304, 432, (n,))], 1).int().cuda()
pairs = mdata.corruption._draw_bar_pairs(v, off, 3)
col = mdata.collate_batches({"input_ids": v}, off, 1024)
notes, lens = col["input_ids"][:, 12:].contiguous(), (col["length"] - 12)


def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = [
    ("masking_token", lambda: mdata.masking_token(v, off, 0.3, u=u), n * 12, lambda i: ob.masking_token(rows[i], np.zeros(len(rows[i])), 0.3)),
    ("masking_note", lambda: mdata.masking_note(v, off, 0.5, u=u), n * 12, lambda i: ob.masking_note(rows[i], np.zeros(len(rows[i])), 0.5)),
    ("randomize_note", lambda: mdata.randomize_note(v, off, 0.5, u=u, new_tokens=new), n * 24,
     lambda i: ob.randomize_note(rows[i], np.zeros(len(rows[i])), np.zeros((len(rows[i]), 3), np.int32), 0.5)),
    ("random_rotating", lambda: mdata.random_rotating(v, off, 3, pairs=pairs), n * 8, lambda i: ob.random_rotating(rows[i], [(0, 1)] * 3)),
    ("collate (1 field)", lambda: mdata.collate_batches({"input_ids": v}, off, 1024), n * 4 + 512 * 1024 * 4, None),
    ("validate_tokens", lambda: mdec.validate_tokens(notes, lens), n * 4, lambda i: ob.validate(rows[i][12:], len(rows[i]) - 12)),
]
print("%d sequences, %d tokens" % (len(rows), n))
for name, fn, nbytes, cpu in cases:
    us = timeit(fn)
    line = "%-18s %8.1f us  %7.1f GB/s algorithmic" % (name, us, nbytes / us / 1e3)
    if cpu is not None:
        t0 = time.perf_counter()
        for i in range(0, 64):
            cpu(i)
        dt = (time.perf_counter() - t0) / 64 * len(rows)
        line += "   | numpy/python oracle, 1 core: %8.1f ms per batch (%.0fx)" % (dt * 1e3, dt * 1e6 / us)
    print(line, flush=True)
