#!/usr/bin/env python3
"""Does the training step gain from running on a pool stream (its keep-bit side stream can then overlap: the default stream shares a
hardware queue pattern with pool streams)?   python tools/debug/train_stream.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
for policy in ("default", "pool", "default", "pool"):
    if policy == "pool":
        with torch.cuda.stream(torch.cuda.Stream()):
            r = bench._time_train(dev, steps=8, warmup=3)
            torch.cuda.synchronize()
    else:
        r = bench._time_train(dev, steps=8, warmup=3)
    print("%-8s %.3f ms per optimizer step" % (policy, r["ms_per_step"]), flush=True)
