#!/usr/bin/env python3
"""Which part of bench.py's kernel-timing phase makes a bert-base-width loop measured AFTER it read 15 - 20 % low?
    python tools/debug/after_profile.py {none|iso|profile|both|alloc}"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda:0")
c2, bb = bench.WORKLOADS["c2"], bench.WORKLOADS["c2-bertbase"]
if what == "iso":
    bench.time_dominant_kernel(c2, "bf16", dev, reps=5, branches=2)
elif what in ("profile", "both"):
    model, diff = bench.build(c2, "bf16", dev, seed=0)
    diff.rng_mode, diff.rng_seed, diff.rng_stream, diff.use_graph = "philox", 105, 0, True
    loop = bench.make_loop(model, diff, c2, "p", dev, 0, 20)
    with torch.no_grad():
        loop.begin()
        for k in range(5):
            loop.advance(k)
        torch.cuda.synchronize()
        rows, span = bench.profile_step(loop, c2, 5, 4, 3.6)
        loop.finish()
    del loop, model, diff
    if what == "both":
        bench.time_dominant_kernel(c2, "bf16", dev, reps=5, branches=2)
elif what == "alloc":
    xs = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(12)]
    del xs
import gc
gc.collect()
torch.cuda.empty_cache()
r = bench._time_loop(bb, "bf16", dev, steps=60, warmup=5)
print("%s -> bert-base width: %.4f ms/step (%.1f steps/s)" % (what, r["ms_per_step"], r["value"]), flush=True)
