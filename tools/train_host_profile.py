#!/usr/bin/env python3
"""Host side of the training step: wall per step, host time to ENQUEUE a step (the GPU kept far behind by a long spin kernel would be ideal;
here: cProfile over a few steps, sorted by own time) - is the step launch-bound?"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from musediffusion_amd import synthetic  # noqa: E402
from musediffusion_amd.train_step import TrainStep  # noqa: E402

dev = torch.device("cuda", 0)
c = bench.WORKLOADS["train"]
model, diff = bench.build(c, "bf16", dev, seed=0)
model.train().requires_grad_(True)
loop = TrainStep(model, diff, microbatch=c["B"], lr=1e-4, weight_decay=0.0, ema_rate=(0.5, 0.9, 0.99), learning_steps=320000,
                 schedule_sampler=bench.make_sampler(os.environ.get("SAMPLER", "uniform"), diff))
cond = synthetic.training_batch(c["B"], c["L"], seed=1)
for _ in range(5):
    loop.run_step(cond)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    loop.run_step(cond)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("10 steps: host enqueue %.2f ms / step, wall %.2f ms / step" % ((t1 - t0) * 100, (t2 - t0) * 100))
# phases
for name, fn in (("forward", None),):
    pass
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(5):
    loop.run_step(cond)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("SORT", "tottime")).print_stats(45)
