#!/usr/bin/env python3
"""The round-6 FFN1 experiment (csrc/gemm_carry.h: the previous tile's GELU epilogue carried under the next tile's K loop, one wave per SIMD)
against the product kernel on FFN1's shape: values (the two sum K in different orders: compared against an fp32 reference of the same
operands) and time, alone on the chip.  M = 32768 is "the full batch" of DESIGN section 5 (two half-batch launches' work)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd import _lib  # noqa: E402

_lib.use_debug_library()
L = _lib.lib()
dev, bf = "cuda", torch.bfloat16
H, F = 512, 2048


def to_panel(x):
    """row-major [rows, cols] -> K32 panel [cols / 32][rows][32]"""
    r, c = x.shape
    return x.reshape(r, c // 32, 32).permute(1, 0, 2).contiguous()


def from_panel(p, rows, cols):
    return p.reshape(cols // 32, rows, 32).permute(1, 0, 2).reshape(rows, cols)


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


for M in [int(v) for v in os.environ.get("MS", "4096,16384,32768").split(",")]:
    torch.manual_seed(M)
    X = (torch.randn(M, H, device=dev)).to(bf)
    W = (torch.randn(F, H, device=dev) / H ** 0.5).to(bf)
    b = torch.randn(F, device=dev) * 0.5
    Xp, Wp = to_panel(X), to_panel(W)
    o_prod, o_carry = torch.zeros(F // 32, M, 32, device=dev, dtype=bf), torch.zeros(F // 32, M, 32, device=dev, dtype=bf)

    def prod():
        _lib.check(L.mh_gemm_bias_act_ex(Xp.data_ptr(), M, 1, Wp.data_ptr(), F, 1, b.data_ptr(), None, 0, 0, o_prod.data_ptr(), M, 1, 0, M, F, H, 2, 1,
                                         _lib.current_stream()))

    def carry(variant=0):
        _lib.check(L.mh_gemm_ffn1_carry(Xp.data_ptr(), M, Wp.data_ptr(), F, b.data_ptr(), o_carry.data_ptr(), M, M, F, H, variant, _lib.current_stream()))

    for variant in (2, 5, 7, 12, 14, 16):
        o_carry.zero_(); carry(variant); torch.cuda.synchronize()
        print("M=%d: variant %d differs from variant 0's outputs: %s" % (M, variant, "checked below" if variant == 0 else ""), end="")
        o_v = o_carry.clone(); o_carry.zero_(); carry(0); torch.cuda.synchronize()
        print(" %d elements" % (o_v != o_carry).sum().item(), flush=True)
    prod(); carry(); torch.cuda.synchronize()
    ref = torch.nn.functional.gelu(X.float() @ W.float().t() + b)
    yp, yc = from_panel(o_prod, M, F).float(), from_panel(o_carry, M, F).float()
    ep, ec = (yp - ref).abs().max().item(), (yc - ref).abs().max().item()
    differ = (yp != yc).float().mean().item()
    print("M=%d: max |error| against fp32: product %.4g, carried %.4g; outputs that differ between the two: %.4f %% (max |diff| %.4g)" %
          (M, ep, ec, 100 * differ, (yp - yc).abs().max().item()), flush=True)
    fl = 2.0 * M * F * H
    rows = (("product 256x128, two blocks per CU", prod), ("carried epilogue, 3-stage ring", carry), ("carried, 6 stages", lambda: carry(2)),
            ("epilogue after its tile", lambda: carry(5)), ("two blocks per CU, epilogue after its tile", lambda: carry(7)),
            ("carried, NO GELU", lambda: carry(11)), ("carried, ORDINARY stores", lambda: carry(12)),
            ("carried, PAIRED full-row stores", lambda: carry(14)), ("two blocks per CU, after its tile, PAIRED", lambda: carry(16)),
            ("carried, PAIRED, NO GELU", lambda: carry(17)), ("two blocks per CU, main loop only", lambda: carry(8)),
            ("main loop only, 3 stages (no epilogue)", lambda: carry(1)), ("product again", prod))
    if os.environ.get("ROWS"):
        rows = [rows[int(k)] for k in os.environ["ROWS"].split(",")]
    for name, fn in rows:
        med, mn = run(fn)
        print("  FFN1 + GELU [%d x %d x %d] %-40s median %6.1f us (min %6.1f) %5.0f TF/s" % (M, F, H, name, med, mn, fl / med / 1e6), flush=True)
