#!/usr/bin/env python3
"""The column-strip kernel (csrc/gemm_strip.h) against gemm_big_kernel's 256 x 128 tile on the launch it serves - FFN1's dense + GELU of K32
panels - alone on the chip (mh_gemm_set_strip), with the QKV scatter on the big tile beside it for scale.  (A QKV form of the strip kernel was
built and measured out in round 6: profiles/r06_ab_nulls.txt.)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd import _lib  # noqa: E402

_lib.use_debug_library()
L = _lib.lib()
dev, bf = "cuda", torch.bfloat16
H, F, SEQ, NH = 512, 2048, 256, 8


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
    X = torch.randn(H // 32, M, 32, device=dev).to(bf)
    W1 = (torch.randn(H // 32, F, 32, device=dev) / H ** 0.5).to(bf)
    Wq = (torch.randn(H // 32, 3 * H, 32, device=dev) / H ** 0.5).to(bf)
    b1, bq = torch.randn(F, device=dev), torch.randn(3 * H, device=dev)
    o1 = torch.empty(F // 32, M, 32, device=dev, dtype=bf)
    q, k, vt = (torch.empty(M * H, device=dev, dtype=bf) for _ in range(3))

    def ffn1():
        _lib.check(L.mh_gemm_bias_act_ex(X.data_ptr(), M, 1, W1.data_ptr(), F, 1, b1.data_ptr(), None, 0, 0, o1.data_ptr(), M, 1, 0, M, F, H, 2, 1, _lib.current_stream()))

    def qkv():
        _lib.check(L.mh_gemm_qkv_vtperm(X.data_ptr(), M, 1, Wq.data_ptr(), 3 * H, 1, bq.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr(), M // SEQ, SEQ, H, NH,
                                        _lib.current_stream()))

    for name, fn, fl in (("FFN1 + GELU [%d x %d x %d]" % (M, F, H), ffn1, 2.0 * M * F * H), ("QKV scatter [%d x %d x %d]" % (M, 3 * H, H), qkv, 2.0 * M * 3 * H * H)):
        for strip in ((0, 1, 0, 1) if fn is ffn1 else (0,)):
            L.mh_gemm_set_strip(strip)
            med, mn = run(fn)
            print("%-36s %-22s median %6.1f us (min %6.1f) %5.0f TF/s" % (name, "strip kernel" if strip else "256x128 big tile", med, mn, fl / med / 1e6), flush=True)
L.mh_gemm_set_strip(1)
