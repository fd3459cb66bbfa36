#!/usr/bin/env python3
"""Launch times of the split-precision GEMM (csrc/split.hip) at config 2's shapes, alone on the chip: the dense + GELU launch against the
same product without the activation and with fp32 instead of split-panel output (what the epilogue and the output bytes cost).
    python tools/split_bench.py [--dtype f16x3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import ops  # noqa: E402
from musediffusion_amd._lib import check, current_stream, lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16x3")
    a = ap.parse_args()
    dt = ops.dtype_code(a.dtype)
    T = torch.bfloat16 if a.dtype == "bf16x3" else torch.float16
    dev = "cuda"
    M = 32768

    def pack(x):
        r, c = x.shape
        out = torch.zeros(2 * (c // 32) * r * 32, dtype=T, device=dev)
        check(lib().mh_split_pack(x.data_ptr(), c, out.data_ptr(), r, r, c, c, dt, current_stream()), "pack")
        return out

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for (N, K, name) in [(2048, 512, "FFN1"), (512, 2048, "FFN2"), (1024, 512, "q|k"), (512, 512, "attention output")]:
        A, W = pack(torch.randn(M, K, device=dev)), pack(torch.randn(N, K, device=dev) * K ** -0.5)
        b = torch.randn(N, device=dev)
        R = pack(torch.randn(M, 512, device=dev)) if N == 512 else None
        g, bt = torch.ones(512, device=dev), torch.zeros(512, device=dev)
        outs = {0: torch.zeros(2 * (N // 32) * M * 32, dtype=T, device=dev), 2: torch.zeros(M, N, device=dev)}
        gf = 3 * 2 * M * N * K / 1e9
        for act in (0, 2) if name == "FFN1" else (0,):
            for mode in (0, 2):
                us = timeit(lambda: check(lib().mh_split_gemm(A.data_ptr(), M, W.data_ptr(), N, b.data_ptr(), 0, None, 0, outs[mode].data_ptr(), M if mode == 0 else N, mode, 0,
                                                              M, N, K, act, dt, current_stream()), "gemm"))
                print("%-17s [%d x %d x 3*%d] act %d out %-12s %7.1f us  %6.0f TFLOP/s issued" % (name, M, N, K, act, "split panels" if mode == 0 else "fp32 rows", us, gf / us * 1e3))
        if N == 512:
            us = timeit(lambda: check(lib().mh_split_gemm_res_ln(A.data_ptr(), M, W.data_ptr(), N, b.data_ptr(), R.data_ptr(), M, g.data_ptr(), bt.data_ptr(), 1e-12,
                                                                 outs[0].data_ptr(), M, M, N, K, dt, current_stream()), "gemm_ln"))
            print("%-17s [%d x %d x 3*%d] + residual + LayerNorm (128 x 512 tile) %7.1f us  %6.0f TFLOP/s issued" % (name, M, N, K, us, gf / us * 1e3))


if __name__ == "__main__":
    main()
