#!/usr/bin/env python3
"""Same-process A/B of training-module switches on the config-5 training step: python tools/ab_train_flags.py NAME [NAME ...]
alternates NAME = True / False over interleaved rounds and prints the median ms per optimizer step of each."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from musediffusion_amd import synthetic, training  # noqa: E402
from musediffusion_amd.train_step import TrainStep  # noqa: E402

dev = torch.device("cuda", 0)
c = bench.WORKLOADS["train"]
model, diff = bench.build(c, "bf16", dev, seed=0)
model.train().requires_grad_(True)
loop = TrainStep(model, diff, microbatch=c["B"], lr=1e-4, weight_decay=0.0, ema_rate=(0.5, 0.9, 0.99), learning_steps=320000)
cond = synthetic.training_batch(c["B"], c["L"], seed=1)


def ms(steps=10):
    for _ in range(3):
        loop.run_step(cond)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        loop.run_step(cond)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for name in sys.argv[1:]:
    res = {True: [], False: []}
    for rnd in range(5):
        for v in (True, False):
            setattr(training, name, v)
            res[v].append(ms())
    setattr(training, name, True)
    print("%s: True %.3f ms  False %.3f ms  (medians of 5 x 10 steps)" % (name, statistics.median(res[True]), statistics.median(res[False])), flush=True)
