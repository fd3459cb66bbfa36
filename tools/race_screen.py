#!/usr/bin/env python3
"""Race screen for the LDS-DMA ring kernels (counted vmcnt + raw barriers are easy to get subtly wrong: a read that passes a
reference check can still race when the DMA happens to land late).  Many launches at varying shapes, every result compared
with a plain fp32 matmul of the same bf16 operands; any wrong tile shows up as an O(1) error."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import _lib, ops  # noqa: E402
_lib.use_debug_library()   # the A/B switches live in libmusehip_dbg.so (include/musehip_dbg.h)

dev = "cuda"
lib = _lib.lib()
g = torch.Generator(device="cpu").manual_seed(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
worst = {}
for it in range(iters):
    M = int(torch.randint(1, 40, (1,), generator=g)) * 128 + int(torch.randint(0, 3, (1,), generator=g)) * 40
    K = [512, 1024, 2048, 96, 3072][it % 5]
    # ---- generic 256x128 tile (+ GELU), QKV scatter, full-row LayerNorm tile (ping-pong), k-major dW
    A = (torch.randn(M, K, generator=g) * 0.5).to(dev).bfloat16()
    for variant in (2, 4):
        lib.mh_gemm_set_variant(variant)
        N = 512
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).bfloat16()
        b = torch.randn(N, generator=g).to(dev) * 0.1
        ref = A.float() @ W.float().T + b
        out = ops.gemm_bias_act(A, W, b, None, None, _lib.MH_BF16)
        e = float((out.float() - ref).abs().max())
        worst["gemm v%d" % variant] = max(worst.get("gemm v%d" % variant, 0.0), e)
        assert e < 0.06, ("generic", variant, it, M, K, e)
    lib.mh_gemm_set_variant(2)
    R = torch.randn(M, 512, generator=g).to(dev).bfloat16()
    gam, bet = (1 + 0.1 * torch.randn(512, generator=g)).to(dev), (0.1 * torch.randn(512, generator=g)).to(dev)
    out = torch.empty(M, 512, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.mh_gemm_bias_res_ln(A.data_ptr(), K, 0, W.data_ptr(), K, 0, b.data_ptr(), R.data_ptr(), 512, 0, gam.data_ptr(), bet.data_ptr(),
                                       1e-12, out.data_ptr(), 512, 0, M, 512, K, _lib.current_stream()))
    ref = torch.nn.functional.layer_norm(A.float() @ W.float().T + b + R.float(), (512,), gam, bet, 1e-12)
    e = float((out.float() - ref).abs().max())
    worst["gemm+ln"] = max(worst.get("gemm+ln", 0.0), e)
    assert e < 0.08, ("ln", it, M, K, e)
    if K % 32 == 0 and M % 32 == 0:
        X = (torch.randn(M, 256, generator=g) * 0.5).to(dev).bfloat16()
        S = int(lib.mh_gemm_dw_splits(M, K, 256))
        part = torch.empty(S, K, 256, device=dev)
        _lib.check(lib.mh_gemm_dw(A.data_ptr(), K, X.data_ptr(), 256, part.data_ptr(), S, M, K, 256, _lib.current_stream()))
        e = float((part.sum(0) - A.float().T @ X.float()).abs().max()) / math.sqrt(M / 1024)
        worst["dw"] = max(worst.get("dw", 0.0), e)
        assert e < 0.05, ("dw", it, M, K, e)
torch.cuda.synchronize()
print("race screen: %d iterations clean; worst errors %s" % (iters, {k: round(v, 4) for k, v in worst.items()}))
