#!/usr/bin/env python3
"""Streaming attention forward with probability dropout at the training shape: in-kernel Philox (writes the keep bits) against a
standalone bit pre-pass (mh_dropout_bits) + the bit-reading forward, and the dropout-free forward."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import _lib  # noqa: E402
_lib.use_debug_library()   # the A/B switches live in libmusehip_dbg.so (include/musehip_dbg.h)
from musediffusion_amd._lib import check, current_stream, lib, ptr  # noqa: E402

B, L, nh, dh = 32, 1024, 8, 64
H = nh * dh
dev = "cuda"
qkv = (torch.randn(B * L, 3 * H, device=dev) * 0.7).to(torch.bfloat16)
vt = torch.zeros(B * nh * dh * L + 256, device=dev, dtype=torch.bfloat16)
L_ = lib()
st = current_stream()
check(L_.mh_head_permute(qkv.data_ptr() + 2 * H * 2, ptr(vt), 3 * H, B, L, nh, dh, 3, 1, st))
out = torch.empty(B * L, H, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B * nh * L, device=dev, dtype=torch.float32)
bits = torch.zeros(int(L_.mh_dropout_bits_words(B * nh, L)), device=dev, dtype=torch.int32)
d = _lib.Dropout(); d.p, d.seed, d.offset, d.mask = 0.1, 42, 7, None


def fwd(drop, bits_in):
    check(L_.mh_attention_stream_fwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * 2, ptr(vt), ptr(out), H, 0, B, L, nh, dh, 1 / math.sqrt(dh),
                                          ptr(lse), L * 3 * H, dh, 3 * H, C.byref(d) if drop else None, ptr(bits) if drop else None, bits_in, st))


def gen():
    check(L_.mh_dropout_bits(ptr(bits), B * nh, L, C.byref(d), st))


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("no dropout            : %7.1f us" % t(lambda: fwd(False, 0)))
print("in-kernel Philox      : %7.1f us" % t(lambda: fwd(True, 0)))
L_.mh_attention_set_stream(3)
print("in-kernel, 16 waves   : %7.1f us" % t(lambda: fwd(True, 0)))
L_.mh_attention_set_stream(1)
print("bit pre-pass          : %7.1f us" % t(gen))
print("bit-reading forward   : %7.1f us" % t(lambda: fwd(True, 1)))
# key-bound compares compiled out (seq_len % 256 == 0) against the general build (mode 4), alternating
for rep in range(3):
    for mode, name in ((1, "full-tile build"), (4, "key-bound build")):
        L_.mh_attention_set_stream(mode)
        print("%s: no dropout %7.1f us   bit-reading %7.1f us" % (name, t(lambda: fwd(False, 0)), t(lambda: fwd(True, 1))))
L_.mh_attention_set_stream(1)

L_.mh_attention_set_stream(2)
print("bit-reading forward, 8 waves x 128-key stages: %7.1f us" % t(lambda: fwd(True, 1)))
L_.mh_attention_set_stream(1)

# ---- backward (dQ kernel + dK/dV kernel in one call), with and without dropout
dctx = (torch.randn(B * L, H, device=dev) * 0.1).to(torch.bfloat16)
dqkv = torch.empty(B * L, 3 * H, device=dev, dtype=torch.bfloat16)
Dv = torch.empty(B * nh * L, device=dev, dtype=torch.float32)
fwd(True, 1)


def bwd(p):
    check(L_.mh_attention_stream_bwd_drop(qkv.data_ptr(), qkv.data_ptr() + H * 2, qkv.data_ptr() + 2 * H * 2, None, None, ptr(dctx), None, ptr(out),
                                          ptr(lse), ptr(Dv), dqkv.data_ptr(), dqkv.data_ptr() + H * 2, dqkv.data_ptr() + 2 * H * 2, 3 * H, B, L, nh, dh,
                                          1 / math.sqrt(dh), L * 3 * H, dh, 3 * H, L * H, dh, H, ptr(bits) if p else None, p, st))


for rep in range(2):
    print("backward (dQ + dK/dV kernels): dropout 0.1 %7.1f us   no dropout %7.1f us" % (t(lambda: bwd(0.1)), t(lambda: bwd(0.0))))
