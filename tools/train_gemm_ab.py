#!/usr/bin/env python3
"""Training-step GEMM shapes (M = 32768 tokens) in the row-major form the tape launches today against the K32-panel form of the sampler:
the measurement the round-6 training refactor is decided on."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import _lib  # noqa: E402

if os.environ.get("VARIANT"):      # VARIANT=4: every launch on the 256x256 tile (debug library's switch)
    _lib.use_debug_library()
    _lib.lib().mh_gemm_set_variant(int(os.environ["VARIANT"]))
L = _lib.lib()
dev, bf = "cuda", torch.bfloat16
M = int(os.environ.get("M", 32768))
H, F = 512, 2048


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


def gx(A, W, b, R, act, N, K, out, pb):
    ap_, wp, op_, rp = pb & 1, (pb >> 1) & 1, (pb >> 2) & 1, (pb >> 3) & 1
    _lib.check(L.mh_gemm_bias_act_ex(A.data_ptr(), M if ap_ else K, ap_, W.data_ptr(), N if wp else K, wp,
                                     b.data_ptr() if b is not None else None, R.data_ptr() if R is not None else None, M if rp else N, rp,
                                     out.data_ptr(), M if op_ else N, op_, 0, M, N, K, {None: 0, "gelu": 2}[act], 1, _lib.current_stream()))


bias = torch.zeros(4096, device=dev)
cases = [("qkv      [M x 1536 x 512]", 1536, 512, None, False), ("ffn1+gelu [M x 2048 x 512]", 2048, 512, "gelu", False),
         ("ffn2+res [M x 512 x 2048]", 512, 2048, None, True), ("ao+res   [M x 512 x 512]", 512, 512, None, True),
         ("dx qkv   [M x 512 x 1536]", 512, 1536, None, True), ("dx ffn2  [M x 2048 x 512] +res", 2048, 512, None, True)]
for name, N, K, act, res in cases:
    A, W, O = t(M, K), t(N, K), torch.empty(M, N, device=dev, dtype=bf)
    R = t(M, N) if res else None
    fl = 2.0 * M * N * K
    row = []
    for pb in (0, 15, 3, 12):
        us = run(lambda: gx(A, W, bias, R, act, N, K, O, pb))
        row.append("panel %2d: %6.1f us %5.0f TF/s" % (pb, us, fl / us / 1e6))
    print("%-34s %s" % (name, "   ".join(row)), flush=True)

# ---- the full-row dense + residual + LayerNorm kernel: sampler form (panel / row-major) against the training form (pre_out, dropout)
import ctypes as C
g_h, b_h = torch.ones(H, device=dev), torch.zeros(H, device=dev)
for name, K in (("ffn2_ln", F), ("ao_ln", H)):
    A, W, R, O, P = t(M, K), t(H, K), t(M, H), torch.empty(M, H, device=dev, dtype=bf), torch.empty(M, H, device=dev, dtype=bf)
    fl = 2.0 * M * H * K
    row = []
    for pb in (0, 15, 3):
        ap_, wp, op_, rp = pb & 1, (pb >> 1) & 1, (pb >> 2) & 1, (pb >> 3) & 1
        us = run(lambda: _lib.check(L.mh_gemm_bias_res_ln(A.data_ptr(), M if ap_ else K, ap_, W.data_ptr(), H if wp else K, wp, b_h.data_ptr(), R.data_ptr(),
                                                           M if rp else H, rp, g_h.data_ptr(), b_h.data_ptr(), 1e-12, O.data_ptr(), M if op_ else H, op_, M, H, K,
                                                           _lib.current_stream())))
        row.append("sampler form panel %2d: %6.1f us %5.0f TF/s" % (pb, us, fl / us / 1e6))
    for p in (0.0, 0.1):
        d = _lib.Dropout()
        d.p, d.seed, d.offset, d.mask = p, 1234, 77, None
        us = run(lambda: _lib.check(L.mh_gemm_bias_dropout_res_ln(A.data_ptr(), K, W.data_ptr(), K, b_h.data_ptr(), R.data_ptr(), H, g_h.data_ptr(), b_h.data_ptr(),
                                                                   1e-12, P.data_ptr(), O.data_ptr(), H, M, H, K, C.byref(d), _lib.current_stream())))
        row.append("training form p=%.1f: %6.1f us %5.0f TF/s" % (p, us, fl / us / 1e6))
    print("%-10s %s" % (name, "   ".join(row)), flush=True)
# ---- FFN1 + GELU with the second output, and the act-grad input gradient
A, W1, O, P = t(M, H), t(F, H), torch.empty(M, F, device=dev, dtype=bf), torch.empty(M, F, device=dev, dtype=bf)
b_f = torch.zeros(F, device=dev)
fl = 2.0 * M * H * F
us1 = run(lambda: _lib.check(L.mh_gemm_bias_act_dact(A.data_ptr(), H, W1.data_ptr(), H, b_f.data_ptr(), P.data_ptr(), O.data_ptr(), F, M, F, H, 2, _lib.current_stream())))
us2 = run(lambda: _lib.check(L.mh_gemm_bias_act_pre(A.data_ptr(), H, W1.data_ptr(), H, b_f.data_ptr(), P.data_ptr(), O.data_ptr(), F, M, F, H, 2, _lib.current_stream())))
print("ffn1 + gelu + dact: %.1f us %.0f TF/s; + pre: %.1f us" % (us1, fl / us1 / 1e6, us2), flush=True)
dY, W2T, dP = t(M, H), t(F, H), torch.empty(M, F, device=dev, dtype=bf)
us = run(lambda: _lib.check(L.mh_gemm_act_grad(dY.data_ptr(), H, W2T.data_ptr(), H, P.data_ptr(), F, dP.data_ptr(), F, M, F, H, 4, _lib.current_stream())))
print("dx ffn2 x dact (act_grad DERIV): %.1f us %.0f TF/s" % (us, fl / us / 1e6), flush=True)
