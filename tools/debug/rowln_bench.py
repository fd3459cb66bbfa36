#!/usr/bin/env python3
"""csrc/gemm_rowln.h against gemm_big_kernel<128x512pp, EPI 3> on the sampler's two LayerNorm GEMMs, alone on the chip (mh_gemm_set_rowln)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd import _lib  # noqa: E402

_lib.use_debug_library()
L = _lib.lib()
dev, bf = "cuda", torch.bfloat16
N = 512


def run(fn, reps=20, rounds=7):
    out = []
    for _ in range(rounds):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(out), min(out)


for M in [int(v) for v in os.environ.get("MS", "16384,32768").split(",")]:
    for K in (512, 2048):
        X = torch.randn(K // 32, M, 32, device=dev).to(bf)
        W = (torch.randn(K // 32, N, 32, device=dev) / K ** 0.5).to(bf)
        R = torch.randn(N // 32, M, 32, device=dev).to(bf)
        b, gam, bet = torch.randn(N, device=dev), torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        o = torch.empty(N // 32, M, 32, device=dev, dtype=bf)

        def fn():
            _lib.check(L.mh_gemm_bias_res_ln(X.data_ptr(), M, 1, W.data_ptr(), N, 1, b.data_ptr(), R.data_ptr(), M, 1, gam.data_ptr(), bet.data_ptr(), 1e-12,
                                             o.data_ptr(), M, 1, M, N, K, _lib.current_stream()))

        for on in (0, 1, 0, 1):
            L.mh_gemm_set_rowln(on)
            med, mn = run(fn)
            print("dense + residual + LayerNorm [%d x %d x %d] %-26s median %6.1f us (min %6.1f) %5.0f TF/s" %
                  (M, N, K, "gemm_rowln_kernel" if on else "128x512 ping-pong tile", med, mn, 2.0 * M * N * K / med / 1e6), flush=True)
L.mh_gemm_set_rowln(1)
