#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at the denoiser's shape."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import _lib, ops  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64); ap.add_argument("--L", type=int, default=512)
ap.add_argument("--nh", type=int, default=8); ap.add_argument("--dh", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev, bf = "cuda", torch.bfloat16
q = torch.randn(a.B, a.nh, a.L, a.dh, device=dev).to(bf)
k = torch.randn(a.B, a.nh, a.L, a.dh, device=dev).to(bf)
vt = torch.randn(a.B * a.nh * a.dh * a.L + 256, device=dev).to(bf)
flops = 4.0 * a.B * a.nh * a.L * a.L * a.dh
for res in (0, 1):
    _lib.lib().mh_attention_set_variant(res)
    ops.attention(q, k, vt, a.dh ** -0.5, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        ops.attention(q, k, vt, a.dh ** -0.5, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print("attention resident=%d: %7.1f us  %6.1f TFLOP/s" % (res, ms * 1e3, flops / ms / 1e9))
