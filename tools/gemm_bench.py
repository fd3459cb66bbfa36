#!/usr/bin/env python3
"""Micro-benchmark of the denoiser's GEMM shapes (used alone and under rocprofv3 --pmc)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import _lib  # noqa: E402
_lib.use_debug_library()   # the A/B switches live in libmusehip_dbg.so (include/musehip_dbg.h)

ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=2)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--zero", action="store_true", help="all-zero operands (clock / power comparison)")
ap.add_argument("--M", type=int, default=32768)
ap.add_argument("--H", type=int, default=512)
ap.add_argument("--F", type=int, default=2048)
ap.add_argument("--shapes", default="ao,ffn1,ffn2,qkv")
ap.add_argument("--dbg", type=int, default=0)
ap.add_argument("--ab", type=str, default="", help="comma list of debug-bit values to alternate between in one process, e.g. 0,32")
ap.add_argument("--noact", action="store_true")
ap.add_argument("--panel", type=int, default=0, help="bit0 A, bit1 W, bit2 out, bit3 residual in K32-panel layout (timing only)")
ap.add_argument("--bufdma", action="store_true", help="A/B the stage DMA as buffer loads (mh_gemm_set_buf_dma) against global_load_lds")
ap.add_argument("--pad", type=int, default=0, help="extra elements on every leading dimension")
a = ap.parse_args()
_lib.lib().mh_gemm_set_variant(a.variant)
_lib.lib().mh_gemm_set_debug(a.dbg)
dev = "cuda"
M, H, F = a.M, a.H, a.F
bf = torch.bfloat16


def t(*s):
    if a.zero:
        return torch.zeros(*s, device=dev, dtype=bf)
    return (torch.randn(*s, device=dev) * (1.0 / s[-1] ** 0.5 if len(s) == 2 and s[0] != M else 1.0)).to(bf)


P = a.pad
X, Xf = t(M, H + P), t(M, F + P)
pb = a.panel


def gx(A, W, b, R, act, N, K, out):
    """timing-only launch with selectable operand layouts (buffers are reinterpreted, values are garbage)"""
    ap_, wp, op_, rp = pb & 1, (pb >> 1) & 1, (pb >> 2) & 1, (pb >> 3) & 1
    _lib.check(_lib.lib().mh_gemm_bias_act_ex(A.data_ptr(), M if ap_ else A.shape[1], ap_, W.data_ptr(), N if wp else W.shape[1], wp,
                                              b.data_ptr(), R.data_ptr() if R is not None else None, M if rp else (R.shape[1] if R is not None else 0), rp,
                                              out.data_ptr(), M if op_ else out.shape[1], op_, 0, M, N, K, {None: 0, "gelu": 2}[act], 1,
                                              _lib.current_stream()))


def gln(A, W, N, K):
    ap_, wp, op_, rp = pb & 1, (pb >> 1) & 1, (pb >> 2) & 1, (pb >> 3) & 1
    _lib.check(_lib.lib().mh_gemm_bias_res_ln(A.data_ptr(), M if ap_ else A.shape[1], ap_, W.data_ptr(), N if wp else W.shape[1], wp,
                                              b_h.data_ptr(), X.data_ptr(), M if rp else X.shape[1], rp, g_h.data_ptr(), b_h.data_ptr(), 1e-12,
                                              o_h.data_ptr(), M if op_ else o_h.shape[1], op_, M, N, K, _lib.current_stream()))


shapes = {
    "ao": lambda: gx(X, Wao, b_h, X, None, H, H, o_h),
    "ffn1": lambda: gx(X, W1, b_f, None, None if a.noact else "gelu", F, H, o_f),
    "ffn2": lambda: gx(Xf, W2, b_h, X, None, H, F, o_h),
    "ao_ln": lambda: gln(X, Wao, H, H),
    "ffn2_ln": lambda: gln(Xf, W2, H, F),
    "ln": lambda: _lib.check(_lib.lib().mh_layernorm_panel(X.data_ptr(), M, g_h.data_ptr(), b_h.data_ptr(), o_h.data_ptr(), M, M, H, 1e-12,
                                                           _lib.current_stream())),
    "qkv": lambda: _lib.check(_lib.lib().mh_gemm_qkv(X.data_ptr(), H, Wqkv.data_ptr(), H, b_q.data_ptr(), q.data_ptr(),
                                                      k.data_ptr(), vt.data_ptr(), M // 512, 512, H, H // 64, 1,
                                                      _lib.current_stream())),
}
Wao, W1, W2, Wqkv = t(H, H + P), t(F, H + P), t(H, F + P), t(3 * H, H)
g_h = torch.ones(H, device=dev)
b_h, b_f, b_q = torch.zeros(H, device=dev), torch.zeros(F, device=dev), torch.zeros(3 * H, device=dev)
o_h, o_f = torch.empty(M, H + P, device=dev, dtype=bf), torch.empty(M, F + P, device=dev, dtype=bf)
q, k, vt = (torch.empty(M * H + 256, device=dev, dtype=bf) for _ in range(3))
flops = {"ao_ln": 2.0 * M * H * H, "ffn2_ln": 2.0 * M * H * F, "ln": 0.0, "ao": 2.0 * M * H * H, "ffn1": 2.0 * M * H * F, "ffn2": 2.0 * M * H * F, "qkv": 2.0 * M * H * 3 * H}
if a.bufdma:
    import statistics
    for name in a.shapes.split(","):
        fn = shapes[name]
        res = {0: [], 1: []}
        for rnd in range(7):
            for v in (0, 1):
                _lib.lib().mh_gemm_set_buf_dma(v)
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record(); torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / a.reps * 1e3)
        print("buf_dma %-8s: global_load_lds %.1f us  buffer_load lds %.1f us" % (name, statistics.median(res[0]), statistics.median(res[1])), flush=True)
    sys.exit(0)
if a.ab:
    import statistics
    vals = [int(v) for v in a.ab.split(",")]
    for name in a.shapes.split(","):
        fn = shapes[name]
        res = {v: [] for v in vals}
        for rnd in range(7):
            for v in vals:
                _lib.lib().mh_gemm_set_debug(v)
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record(); torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / a.reps * 1e3)
        print("A/B %-8s variant %d: " % (name, a.variant) + "  ".join("dbg %d: median %.1f us (min %.1f)" % (v, statistics.median(r), min(r)) for v, r in res.items()), flush=True)
    sys.exit(0)
for name in a.shapes.split(","):
    fn = shapes[name]
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    if a.dbg & 16:
        nw = 4 if a.variant == 2 and a.panel else 8
        pr = o_f.view(torch.int32).flatten()[64:64 + 8 * nw * 1024].view(-1, 8).cpu().double()
        pr = pr[pr[:, 3] == 16]
        print("  prof over %d waves: wait %.0f  barrier %.0f  work %.0f shader clocks per K-step; loop %.0f clocks, in-kernel clock %.2f GHz" % (
            pr.shape[0], pr[:, 0].mean() / 15, pr[:, 1].mean() / 15, pr[:, 2].mean() / 14, pr[:, 4].mean(),
            (pr[:, 4] / pr[:, 5]).median() * 0.1))
    print("panel %d dbg %d %-5s variant %d: %8.1f us  %7.1f TFLOP/s" % (a.panel, a.dbg, name, a.variant, ms * 1e3, flops[name] / ms / 1e9), flush=True)
