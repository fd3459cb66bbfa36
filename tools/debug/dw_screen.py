#!/usr/bin/env python3
"""Repeated-launch screen of the weight-gradient GEMM (mh_gemm_dw_bias + mh_sum_slices).  Launches ALTERNATE between two operand
sets (a block that read its LDS before the DMA landed would otherwise find the same bytes there from the launch before) and every
result is bit-compared with the first result of its own set: the kernels are deterministic, any difference is a race."""
import sys, os, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd._lib import check, current_stream, lib

dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for (K, M, N) in [(8192, 512, 2048), (8192, 2048, 512), (32768, 512, 512), (32768, 1536, 512)]:
    g = torch.Generator().manual_seed(K + M + N)
    sets = []
    for _ in range(2):
        A = (torch.randn(K, M, generator=g) * 0.5).to(dev).bfloat16().contiguous()
        B = (torch.randn(K, N, generator=g) * 0.5).to(dev).bfloat16().contiguous()
        sets.append([A, B, A.float().T @ B.float(), A.float().sum(0), None])
    S = int(lib().mh_gemm_dw_splits(K, M, N))
    n = M * N + M
    bad, worst = 0, 0.0
    for it in range(iters):
        A, B, ref, ref_b, first = sets[it & 1]
        part = torch.full((S, n), float("nan"), device=dev)
        out = torch.empty(n, device=dev)
        check(lib().mh_gemm_dw_bias(A.data_ptr(), M, B.data_ptr(), N, part.data_ptr(), S, K, M, N, 1, current_stream()))
        check(lib().mh_sum_slices(part.data_ptr(), S, n, out.data_ptr(), current_stream()))
        if first is None:
            sets[it & 1][4] = out.clone()
            err = (out[:M * N].view(M, N) - ref).abs().max().item()
            errb = (out[M * N:] - ref_b).abs().max().item()
            print("K=%d M=%d N=%d splits=%d: max err vs fp32 matmul %.3e (dW) %.3e (db)" % (K, M, N, S, err, errb), flush=True)
        elif not torch.equal(out, first):
            d = (out - first).abs()
            bad += 1
            worst = max(worst, d.max().item())
            if bad <= 3:
                idx = int(d.argmax())
                print("  iteration %d differs from the first launch: %d elements, max %.3e at flat %d" % (it, int((d > 0).sum()), d.max().item(), idx), flush=True)
    print("  %d / %d launches differ from the first (worst %.3e)" % (bad, iters - 1, worst), flush=True)
