#!/usr/bin/env python3
"""What the deferred-LayerNorm epilogues cost at d_model 768 (the reference-true width has no full-row tile): the attention-output dense
[16384 x 768 x 768] bare, writing the output rows' statistics (DO), normalising its residual on the fly (DR), and both (the form the step
runs); FFN1 [16384 x 3072 x 768] bare against raw A rows + statistics (DA)."""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd import _lib  # noqa: E402

L = _lib.lib()
dev, bf = "cuda", torch.bfloat16
M, H, F = int(os.environ.get("M", 16384)), 768, 3072
S = (H + 127) // 128


def t(*s):
    return (torch.randn(*s, device=dev) * 0.05).to(bf)


def run(fn, reps=20, rounds=5):
    out = []
    for _ in range(rounds):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(out)


A, W, R, O = t(M * H), t(H * H), t(M * H), torch.empty(M * H, device=dev, dtype=bf)
bias, gam, bet, c1 = (torch.randn(F, device=dev) * 0.1 for _ in range(4))
st_in, st_out = torch.rand(M, S, 2, device=dev) + 1.0, torch.empty(M, S, 2, device=dev)


def ao(do, dr):
    d = _lib.LnDefer()
    d.h_norm, d.eps = H, 1e-12
    if dr:
        d.r_stats, d.r_slots, d.r_gamma, d.r_beta = st_in.data_ptr(), S, gam.data_ptr(), bet.data_ptr()
    if do:
        d.o_stats, d.o_slots = st_out.data_ptr(), S
    _lib.check(L.mh_gemm_bias_act_defer(A.data_ptr(), M, W.data_ptr(), H, bias.data_ptr(), R.data_ptr(), M, O.data_ptr(), M, M, H, H, 0,
                                        C.byref(d) if (do or dr) else None, _lib.current_stream()))


fl = 2.0 * M * H * H
for name, do, dr in (("bare (+ residual)", 0, 0), ("+ output statistics (DO)", 1, 0), ("+ raw residual (DR)", 0, 1), ("DO + DR (as the step)", 1, 1)):
    us = run(lambda: ao(do, dr))
    print("attention output [%d x %d x %d] %-28s %6.1f us %5.0f TF/s" % (M, H, H, name, us, fl / us / 1e6), flush=True)
W1, Of = t(F * H), torch.empty(M * F, device=dev, dtype=bf)


def ffn1(da):
    d = _lib.LnDefer()
    d.h_norm, d.eps = H, 1e-12
    if da:
        d.a_stats, d.a_slots, d.c1 = st_in.data_ptr(), S, c1.data_ptr()
    _lib.check(L.mh_gemm_bias_act_defer(A.data_ptr(), M, W1.data_ptr(), F, bias.data_ptr(), None, 0, Of.data_ptr(), M, M, F, H, 2,
                                        C.byref(d) if da else None, _lib.current_stream()))


fl = 2.0 * M * F * H
for name, da in (("bare", 0), ("raw A rows (DA)", 1)):
    us = run(lambda: ffn1(da))
    print("FFN1 + GELU [%d x %d x %d] %-28s %6.1f us %5.0f TF/s" % (M, F, H, name, us, fl / us / 1e6), flush=True)
