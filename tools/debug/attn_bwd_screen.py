#!/usr/bin/env python3
"""Repeated-launch screen of the attention backward kernels (LDS-DMA double buffer + transposing reads of the same stages + keep words
staged with the operands): N launches on the same operands, dq / dk / dv compared bit for bit with the first launch, with and without
dropout, at a whole-stage and a ragged sequence length."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd import _lib  # noqa: E402
from musediffusion_amd._lib import check, current_stream, lib, ptr  # noqa: E402

import ctypes as C  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda"
L_, st = lib(), current_stream()
for B, L, nh, dh in ((8, 1024, 8, 64), (3, 528, 4, 64), (4, 512, 4, 32)):
    H = nh * dh
    g = torch.Generator().manual_seed(L)
    qkv = (torch.randn(B * L, 3 * H, generator=g) * 0.7).to(dev).bfloat16()
    vt = torch.zeros(B * nh * dh * L + 256, device=dev, dtype=torch.bfloat16)
    check(L_.mh_head_permute(qkv.data_ptr() + 2 * H * 2, ptr(vt), 3 * H, B, L, nh, dh, 3, 1, st))
    out = torch.empty(B * L, H, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * nh * L, device=dev, dtype=torch.float32)
    bits = torch.zeros(int(L_.mh_dropout_bits_words(B * nh, L)), device=dev, dtype=torch.int32)
    d = _lib.Dropout(); d.p, d.seed, d.offset, d.mask = 0.1, 42, 7, None
    dctx = (torch.randn(B * L, H, generator=g) * 0.1).to(dev).bfloat16()
    Dv = torch.empty(B * nh * L, device=dev, dtype=torch.float32)
    for p in (0.1, 0.0):
        check(L_.mh_attention_stream_fwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * 2, ptr(vt), ptr(out), H, 0, B, L, nh, dh, 1 / math.sqrt(dh), ptr(lse),
                                              L * 3 * H, dh, 3 * H, C.byref(d) if p else None, ptr(bits) if p else None, 0, st))
        first, bad = None, 0
        for it in range(N):
            dqkv = torch.full((B * L, 3 * H), float("nan"), device=dev, dtype=torch.bfloat16)
            check(L_.mh_attention_stream_bwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * 2, qkv.data_ptr() + 2 * H * 2, None, None, ptr(dctx), None, ptr(out),
                                                  ptr(lse), ptr(Dv), dqkv.data_ptr(), dqkv.data_ptr() + H * 2, dqkv.data_ptr() + 2 * H * 2, 3 * H, B, L, nh,
                                                  dh, 1 / math.sqrt(dh), L * 3 * H, dh, 3 * H, L * H, dh, H, ptr(bits) if p else None, p, st))
            if first is None:
                first = dqkv.clone()
                assert not torch.isnan(first.float()).any()
            elif not torch.equal(first, dqkv):
                bad += 1
        print("attention backward B=%d L=%d nh=%d dh=%d dropout %.1f: %d / %d launches differ from the first" % (B, L, nh, dh, p, bad, N - 1))
        assert bad == 0
