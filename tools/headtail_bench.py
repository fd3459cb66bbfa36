#!/usr/bin/env python3
"""Stand-alone timing of the fused head / tail kernels of the denoiser (csrc/headtail.hip) at BASELINE config 2's half batch
(16384 rows, E 128, d_model 512, vocabulary 729): up-projection + LayerNorm; down-projection alone; + rounding; + rounding + update.
    python tools/headtail_bench.py [--rows 16384] [--reps 50]"""
import argparse
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=16384)
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
dev = "cuda"
N, E, H, V, L = a.rows, 128, 512, 729, 512
B = N // L
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
panel = lambda w: w.bfloat16().reshape(w.shape[0], w.shape[1] // 32, 32).permute(1, 0, 2).contiguous()
x = rnd(N, E, sc=0.5)
w0, b0, w2, b2 = panel(rnd(H, E, sc=1 / math.sqrt(E))), rnd(H, sc=0.1), panel(rnd(H, H, sc=1 / math.sqrt(H))), rnd(H, sc=0.1)
pos, emb, gam, bet = rnd(L, H, sc=0.5), rnd(B, H, sc=0.5), 1 + rnd(H, sc=0.1), rnd(H, sc=0.1)
X = torch.empty(H // 32, N, 32, device=dev, dtype=torch.bfloat16)
d0, db0, d2, db2 = panel(rnd(H, H, sc=1 / math.sqrt(H))), rnd(H, sc=0.1), panel(rnd(E, H, sc=1 / math.sqrt(H))), rnd(E, sc=0.1)
out, sq, idx = torch.empty(N, E, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev, dtype=torch.int32)
table = rnd(V, E)
buf = torch.empty(int(lib.mh_round_split_bytes(E, V)), dtype=torch.uint8, device=dev)
st = _lib.current_stream()
_lib.check(lib.mh_round_split_table(table.data_ptr(), None, V, E, buf.data_ptr(), st))
xt, xs, pred = rnd(N, E), rnd(N, E), torch.empty(N, E, device=dev)
mask = (torch.rand(N, generator=g) > 0.1).to(torch.int32).to(dev)
coef = torch.tensor([0.3, 0.7, 0.1, 1.0, 0.5, 0.9, 0.2, 0.0], device=dev)
ctr = torch.zeros(1, dtype=torch.int32, device=dev)
rng = _lib.StepRng(); rng.seed, rng.stream_id, rng.bound, rng.step_counter, rng.first_elem = 105, 0, 1.0, ctr.data_ptr(), 0
upd = _lib.StepUpdate()
upd.x, upd.x_start, upd.mask, upd.mask_per_elem, upd.table, upd.coef = xt.data_ptr(), xs.data_ptr(), mask.data_ptr(), 0, table.data_ptr(), coef.data_ptr()
upd.clip, upd.ddim, upd.pred_xstart, upd.mean_out, upd.noise, upd.rng = 1, 0, pred.data_ptr(), None, None, C.pointer(rng)


def head():
    _lib.check(lib.mh_up_proj_ln_fused(x.data_ptr(), E, E, w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), b2.data_ptr(), pos.data_ptr(), emb.data_ptr(), None,
                                       gam.data_ptr(), bet.data_ptr(), 1e-12, X.data_ptr(), N, B, L, H, st))


def tail(kind):
    if kind == 0:
        _lib.check(lib.mh_down_proj_fused(X.data_ptr(), N, d0.data_ptr(), db0.data_ptr(), d2.data_ptr(), db2.data_ptr(), out.data_ptr(), sq.data_ptr(), N, E, H, st))
    else:
        _lib.check(lib.mh_down_proj_round_fused(X.data_ptr(), N, d0.data_ptr(), db0.data_ptr(), d2.data_ptr(), db2.data_ptr(), out.data_ptr(), None, buf.data_ptr(), V,
                                                idx.data_ptr(), C.byref(upd) if kind == 2 else None, N, E, H, st))


def timeit(fn, name):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-44s %7.2f us per launch" % (name, e0.elapsed_time(e1) / a.reps * 1e3), flush=True)


head()
timeit(head, "head: up-projection + pos/time + LayerNorm")
timeit(lambda: tail(0), "tail: down-projection (+ |row|^2)")
timeit(lambda: tail(1), "tail: down-projection + rounding")
timeit(lambda: tail(2), "tail: down-projection + rounding + update")
