#!/usr/bin/env python3
"""Practical-ceiling yardstick for the denoiser's dense layers: the vendor library (torch.matmul on bf16 ->
hipBLASLt / rocBLAS) next to libmusehip's kernels on the shapes of one reverse step, same box, same random operands.

TOOLS ONLY: the vendor GEMM is never on the product path; it is here to say what a dense bf16 product of these
shapes can reach on this part, so that each hand-written kernel has a measured target instead of the 2.5 PF datasheet figure.

For every shape [M x N x K] (out[M, N] = X[M, K] W[N, K]^T, the nn.Linear convention) and for M = a batch slice (16384) and the full
batch (32768):
  lib       torch.matmul(X, W.T)                                   - plain product, bf16 output, no bias / activation
  ours      mh_gemm_bias_act_ex on K32-panel operands (what the engine launches), bias, no activation, bf16 output
  ours+epi  the launch as it runs in the step (FFN1: + GELU; dense + residual + LayerNorm for the N = d_model products)
"warm" = the same operands back to back (weights and activations L2 / Infinity-Cache resident as far as they fit);
"cold" = rotating over enough operand sets that no launch re-reads what an earlier one left in the 256 MB Infinity Cache.
Rounds of (lib, ours, ours+epi) are interleaved in ONE process; medians are printed (cdna_hip_programming.md rule 24).

    python tools/gemm_yardstick.py [--width 512|768] [--reps 20] [--rounds 7]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=512, help="d_model: 512 (BASELINE config 2) or 768 (bert-base, the reference-true width)")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--sets", type=int, default=6, help="operand sets of the cold variant")
ap.add_argument("--only", default="", help="comma list of variants to run (lib, ours, ours+epi); under rocprofv3 `--only lib` names the vendor kernels")
a = ap.parse_args()
dev, bf = "cuda", torch.bfloat16
H = a.width
F = 4 * H
L = _lib.lib()
S = _lib.current_stream


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(bf)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def shape_case(name, M, N, K, act, ln):
    """returns {variant: callable(i)} over `sets` operand sets (index i % sets); warm = sets 1"""
    nset = a.sets
    X = [rnd(M, K) for _ in range(nset)]
    W = rnd(N, K, scale=K ** -0.5)
    bias = torch.zeros(N, device=dev)
    out = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(nset)]
    res = rnd(M, N) if ln else None
    gam, bet = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    Wt = W.t()

    def lib(i=0, n=1):
        torch.matmul(X[i % n], Wt, out=out[i % n])

    # panel flags 1: the buffers are reinterpreted as K32 panels (timing only: same bytes, same access pattern as in the step)
    def ours(i=0, n=1):
        _lib.check(L.mh_gemm_bias_act_ex(X[i % n].data_ptr(), M, 1, W.data_ptr(), N, 1, bias.data_ptr(), None, 0, 0, out[i % n].data_ptr(), M, 1, 0,
                                         M, N, K, 0, 1, S()))

    def ours_epi(i=0, n=1):
        if ln:
            _lib.check(L.mh_gemm_bias_res_ln(X[i % n].data_ptr(), M, 1, W.data_ptr(), N, 1, bias.data_ptr(), res.data_ptr(), M, 1, gam.data_ptr(),
                                             bet.data_ptr(), 1e-12, out[i % n].data_ptr(), M, 1, M, N, K, S()))
        else:
            _lib.check(L.mh_gemm_bias_act_ex(X[i % n].data_ptr(), M, 1, W.data_ptr(), N, 1, bias.data_ptr(), None, 0, 0, out[i % n].data_ptr(), M, 1, 0,
                                             M, N, K, 2 if act else 0, 1, S()))
    v = {"lib": lib, "ours": ours}
    if act or (ln and L.mh_gemm_bias_res_ln_supported(N)):
        v["ours+epi"] = ours_epi
    if a.only:
        v = {k: f for k, f in v.items() if k in a.only.split(",")}
    return v


shapes = [("ffn1", F, H, True, False), ("ffn2", H, F, False, True), ("qkv", 3 * H, H, False, False), ("ao", H, H, False, True)]
print("# d_model %d, ffn %d; us per launch (TFLOP/s), medians over %d interleaved rounds of %d launches" % (H, F, a.rounds, a.reps))
for M in (16384, 32768):
    for name, N, K, act, ln in shapes:
        v = shape_case(name, M, N, K, act, ln)
        flops = 2.0 * M * N * K
        for mode, n in (("warm", 1), ("cold", a.sets)):
            res = {k: [] for k in v}
            for _ in range(a.rounds):
                for k, fn in v.items():
                    res[k].append(timed(lambda i=0, fn=fn: fn(i, n), a.reps))
            line = "  ".join("%-8s %7.1f us (%6.0f)" % (k, statistics.median(r), flops / statistics.median(r) / 1e6) for k, r in res.items())
            print("%-5s [%5d x %4d x %4d] %s: %s" % (name, M, N, K, mode, line), flush=True)
        del v
        torch.cuda.empty_cache()
