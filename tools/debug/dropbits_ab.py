#!/usr/bin/env python3
"""Bit-identity of mh_dropout_bits between two builds of the library (the keep rule was re-expressed, not changed) + timing.
    python tools/debug/dropbits_ab.py tools/ab/libmusehip_base.so"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from musediffusion_amd import _lib  # noqa: E402


class mh_dropout(ctypes.Structure):
    _fields_ = [("p", ctypes.c_float), ("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64), ("mask", ctypes.c_void_p)]


new = _lib.lib()
old = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for L_ in (new, old):
    L_.mh_dropout_bits.restype = ctypes.c_int
    L_.mh_dropout_bits.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(mh_dropout), ctypes.c_void_p]
    L_.mh_dropout_bits_words.restype = ctypes.c_size_t
    L_.mh_dropout_bits_words.argtypes = [ctypes.c_int, ctypes.c_int]
stream = torch.cuda.current_stream().cuda_stream
for BH, L, p in ((256, 1024, 0.1), (6, 528, 0.1), (4, 100, 0.37), (3, 64, 0.999), (5, 96, 0.00390625), (2, 2096, 0.25)):
    n = new.mh_dropout_bits_words(BH, L)
    d = mh_dropout(p, 0x1234567887654321, (7 << 16) | 3, None)
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    b = torch.zeros(n, dtype=torch.int32, device="cuda")
    assert new.mh_dropout_bits(a.data_ptr(), BH, L, ctypes.byref(d), stream) == 0
    assert old.mh_dropout_bits(b.data_ptr(), BH, L, ctypes.byref(d), stream) == 0
    torch.cuda.synchronize()
    same = bool(torch.equal(a, b))
    t = {}
    for name, lib in (("new", new), ("old", old)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.mh_dropout_bits(a.data_ptr(), BH, L, ctypes.byref(d), stream)
        e1.record()
        torch.cuda.synchronize()
        t[name] = e0.elapsed_time(e1) * 100
    print("BH=%d L=%d p=%g: identical=%s  new %.1f us  old %.1f us" % (BH, L, p, same, t["new"], t["old"]))
    assert same
