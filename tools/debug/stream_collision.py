#!/usr/bin/env python3
"""Does the pair of batch-slice chains lose its concurrency when the process already owns other streams?  HIP maps streams onto a few
hardware queues; two chains that land on ONE queue run back to back.  Creates k extra streams (each used once) before the loop is
built and times config 2.      python tools/debug/stream_collision.py [--workload c2]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2")
ap.add_argument("--ks", default="0,1,2,3,4,5,6,7,8")
a = ap.parse_args()
dev = torch.device("cuda:0")
c = bench.WORKLOADS[a.workload]
keep = []
for k in [int(v) for v in a.ks.split(",")]:
    while len(keep) < k:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(16, device=dev).add_(1)
        keep.append(s)
    torch.cuda.synchronize()
    r = bench._time_loop(c, "bf16", dev, steps=60, warmup=5)
    print("extra streams %d: %.4f ms/step (%.1f steps/s)" % (len(keep), r["ms_per_step"], r["value"]), flush=True)
