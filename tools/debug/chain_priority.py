#!/usr/bin/env python3
"""A/B: one of the two decoupled chains on a high-priority stream (its workgroups are dispatched first)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from musediffusion_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
c = bench.WORKLOADS["c2"]
orig = ops.concurrent_streams


def with_priority(n, device=None, tries=8):
    for _ in range(tries):
        ss = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device, priority=-1)]
        if ops.streams_overlap(ss[0], ss[1]):
            return ss
    return ss


res = {"equal": [], "one high": []}
for rnd in range(5):
    for name, fn in (("equal", orig), ("one high", with_priority)):
        ops.concurrent_streams = fn
        r = bench._time_loop(c, "bf16", dev, steps=60, warmup=5)
        res[name].append(r["ms_per_step"])
for k, v in res.items():
    print("%-9s" % k, " ".join("%.3f" % t for t in v), flush=True)
