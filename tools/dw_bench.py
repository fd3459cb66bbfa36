#!/usr/bin/env python3
"""Launch times of the weight-gradient GEMM (mh_gemm_dw_bias: dW = dY^T X over the token range, split-K partials + column sums) at the training step's
shapes, alone on the chip.  Two library builds are compared by running it once per build (MUSEHIP_AB=1 MUSEHIP_LIB=... selects the other one), alternated.
    python tools/dw_bench.py [--reps 30] [--rounds 5]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd._lib import check, current_stream, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = "cuda"
K = 32768
for (M, N) in [(2048, 512), (512, 2048), (512, 512), (1536, 512), (512, 128)]:
    A = (torch.randn(K, M, device=dev) * 0.5).bfloat16()
    B = (torch.randn(K, N, device=dev) * 0.5).bfloat16()
    S = int(lib().mh_gemm_dw_splits(K, M, N))
    part = torch.empty(S, M * N + M, device=dev)

    def fn():
        check(lib().mh_gemm_dw_bias(A.data_ptr(), M, B.data_ptr(), N, part.data_ptr(), S, K, M, N, 1, current_stream()))
    ts = []
    for _ in range(a.rounds):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / a.reps * 1e3)
    t = statistics.median(ts)
    print("dW [%4d x %4d] over %d tokens, %3d splits: %7.1f us  %6.0f TFLOP/s" % (M, N, K, S, t, 2.0 * K * M * N / t / 1e6), flush=True)
