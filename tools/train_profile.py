#!/usr/bin/env python3
"""Per-kernel time of one training micro-batch (training_losses fwd + bwd, config 5 shape) from the library's own per-launch HIP
events (mh_profile_start / mh_profile_stop).  Usage: python tools/train_profile.py [--dropout 0.1] [--steps 2]"""
import argparse
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from musediffusion_amd import _lib, synthetic  # noqa: E402
_lib.use_debug_library()   # the A/B switches live in libmusehip_dbg.so (include/musehip_dbg.h)

ap = argparse.ArgumentParser()
ap.add_argument("--dropout", type=float, default=0.1)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--dw-blocks", type=int, default=0, help="A/B: blocks a weight-gradient launch aims for (mh_gemm_dw_set_blocks)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
if a.dw_blocks:
    _lib.check(_lib.lib().mh_gemm_dw_set_blocks(a.dw_blocks))
c = bench.WORKLOADS["train"]
model, diff = bench.build(c, "bf16", dev, seed=0)
model.dropout.p = a.dropout
model.bert_hidden_dropout = model.bert_attention_dropout = a.dropout
model.train().requires_grad_(True)
batch = {k: v.to(dev) for k, v in synthetic.training_batch(c["B"], c["L"], seed=1).items()}
g = torch.Generator().manual_seed(7)


def step():
    t = torch.randint(0, c["T"], (c["B"],), generator=g).to(dev)
    model.zero_grad(set_to_none=True)
    terms = diff.training_losses(model, t, model_kwargs=batch)
    terms["loss"].mean().backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    step()
e1.record()
torch.cuda.synchronize()
print("dropout %.2f: %.2f ms per micro-batch (fwd + bwd, unprofiled)" % (a.dropout, e0.elapsed_time(e1) / 3))
lib = _lib.lib()
_lib.check(lib.mh_profile_start())
for _ in range(a.steps):
    step()
buf = ctypes.create_string_buffer(1 << 24)
need = lib.mh_profile_stop(buf, len(buf))
assert 0 < need <= len(buf)
acc = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for line in buf.value.decode().splitlines():
    kernel, note, grid, block, stream, ms = line.split("\t")
    key = kernel.strip("()")[:60] + (" | " + " ".join(x for x in note.split() if x.split("=")[0] in ("tile", "epi", "act", "M", "N", "K", "drop", "splits")) if note else "")
    acc[key][0] += 1
    acc[key][1] += float(ms)
    tot += float(ms)
print("%-110s %6s %9s %6s" % ("kernel | shape", "n/step", "ms/step", "share"))
for k, (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:32]:
    print("%-110s %6.1f %9.3f %5.1f%%" % (k[:110], n / a.steps, ms / a.steps, 100 * ms / tot))
print("sum of kernel spans: %.2f ms per step" % (tot / a.steps))
