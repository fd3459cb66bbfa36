#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at the denoiser's shape."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musediffusion_amd import _lib, ops  # noqa: E402
_lib.use_debug_library()   # the A/B switches live in libmusehip_dbg.so (include/musehip_dbg.h)
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64); ap.add_argument("--L", type=int, default=512)
ap.add_argument("--nh", type=int, default=8); ap.add_argument("--dh", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--prof", action="store_true", help="only the in-kernel phase profile of the streaming forward (ablation 128)")
a = ap.parse_args()
dev, bf = "cuda", torch.bfloat16
q = torch.randn(a.B, a.nh, a.L, a.dh, device=dev).to(bf)
k = torch.randn(a.B, a.nh, a.L, a.dh, device=dev).to(bf)
vt = torch.randn(a.B * a.nh * a.dh * a.L + 256, device=dev).to(bf)
flops = 4.0 * a.B * a.nh * a.L * a.L * a.dh
if a.prof:
    # ablation 128: every wave sums, over its stages, the clocks it spends waiting for the stage's DMA, at the barrier behind it, in the
    # stage's work and at the end-of-stage barrier, and leaves the four sums in the (otherwise unused) keep_bits buffer
    ctx = torch.empty(a.B * a.L, a.nh * a.dh, device=dev, dtype=bf)
    nblk = min(a.B * a.nh * ((a.L + 511) // 512), 256)
    buf = torch.zeros(nblk * 16 * 4, device=dev, dtype=torch.int64)
    sB, sH = a.nh * a.L * a.dh, a.L * a.dh
    _lib.lib().mh_attention_set_ablation(128)
    for rep in range(3):
        buf.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().mh_attention_stream_fwd_drop(q.data_ptr(), k.data_ptr(), vt.data_ptr(), ctx.data_ptr(), a.nh * a.dh, 0, a.B, a.L, a.nh, a.dh,
                                                           a.dh ** -0.5, None, sB, sH, a.dh, None, buf.data_ptr(), 0, _lib.current_stream()))
        e1.record(); torch.cuda.synchronize()
        t = buf.view(nblk, 16, 4).double().cpu()
        tot = t.sum(-1)
        print("launch %.1f us; per wave (shader clocks, mean over %d blocks): DMA wait %.0f  top barrier %.0f  stage work %.0f  end barrier %.0f  (sum %.0f)"
              % (e0.elapsed_time(e1) * 1e3, nblk, *[float(t[:, :, i].mean()) for i in range(4)], float(tot.mean())))
        print("  by wave index: work " + " ".join("%5.0f" % float(t[:, w, 2].mean()) for w in range(16)))
        print("  by wave index: barriers " + " ".join("%5.0f" % float((t[:, w, 1] + t[:, w, 3]).mean()) for w in range(16)))
        print("  blocks: work min %.0f max %.0f; barrier wait min %.0f max %.0f" % (float(t[:, :, 2].min()), float(t[:, :, 2].max()),
              float((t[:, :, 1] + t[:, :, 3]).min()), float((t[:, :, 1] + t[:, :, 3]).max())))
    _lib.lib().mh_attention_set_ablation(0)
    raise SystemExit(0)
for res in (0, 1, 2):
    _lib.lib().mh_attention_set_variant(res)
    ops.attention(q, k, vt, a.dh ** -0.5, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        ops.attention(q, k, vt, a.dh ** -0.5, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print("attention resident=%d: %7.1f us  %6.1f TFLOP/s" % (res, ms * 1e3, flops / ms / 1e9))

if _lib.lib().mh_attention_stream_supported(a.L, a.dh):
    ctx = torch.empty(a.B * a.L, a.nh * a.dh, device=dev, dtype=bf)
    def run():
        _lib.check(_lib.lib().mh_attention_stream_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), ctx.data_ptr(), a.nh * a.dh, 0, a.B, a.L,
                                                      a.nh, a.dh, a.dh ** -0.5, _lib.current_stream()))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print("attention stream    : %7.1f us  %6.1f TFLOP/s" % (ms * 1e3, flops / ms / 1e9))
if _lib.lib().mh_attention_stream_prescaled_supported(a.L, a.dh):
    ctx = torch.empty(a.B * a.L, a.nh * a.dh, device=dev, dtype=bf)
    qp = (q.float() * (a.dh ** -0.5 * 1.4426950408889634)).to(bf)
    def run_pre():
        _lib.check(_lib.lib().mh_attention_stream_fwd_prescaled(qp.data_ptr(), k.data_ptr(), vt.data_ptr(), ctx.data_ptr(), a.nh * a.dh, 0, a.B, a.L,
                                                                a.nh, a.dh, _lib.current_stream()))
    for name, fn in (("stream", run), ("prescaled", run_pre), ("stream", run), ("prescaled", run_pre)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps * 5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (a.reps * 5)
        print("attention %-9s: %7.1f us  %6.1f TFLOP/s" % (name, ms * 1e3, flops / ms / 1e9))
if a.L <= 512 and a.dh == 64:
    names = {0: "everything", 1: "no softmax VALU", 2: "no S MFMAs", 4: "no PV MFMAs", 6: "no MFMAs", 7: "no MFMAs, no softmax", 8: "no LDS reads",
             16: "no DMA", 24: "no DMA, no LDS reads", 31: "nothing (Q load, loop skeleton, stores)", 32: "everything but the context stores",
             63: "skeleton without the stores", 95: "skeleton without the Q loads", 127: "skeleton without Q loads and stores"}
    for rnd in range(2):
        for abl in (0, 1, 2, 4, 6, 7, 8, 16, 24, 31, 32, 63, 95, 127):
            _lib.lib().mh_attention_set_ablation(abl)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps * 5):
                run()
            e1.record(); torch.cuda.synchronize()
            print("ablation %2d %-42s: %7.1f us per launch" % (abl, names[abl], e0.elapsed_time(e1) / (a.reps * 5) * 1e3))
    _lib.lib().mh_attention_set_ablation(0)
# per-block timeline of the resident kernel (100 MHz stamps)
for res in (1, 2):
    _lib.lib().mh_attention_set_variant(res)
    st = torch.zeros(a.B * a.nh * 32, dtype=torch.int64, device=dev)
    _lib.lib().mh_attention_set_profile(st.data_ptr())
    ops.attention(q, k, vt, a.dh ** -0.5, 1)
    torch.cuda.synchronize()
    _lib.lib().mh_attention_set_profile(None)
    s_ = st.view(-1, 32).cpu().double()
    nw = 8 if res == 1 else 16
    t0 = s_[:, 0].min()
    stage = (s_[:, 1] - s_[:, 0]).mean() / 100
    ends = s_[:, 2:2 + nw]
    print("resident=%d: staging %.2f us/block; block duration %.2f us (first wave done after %.2f, last %.2f); blocks start at %s us" % (
        res, stage, ((ends.max(1).values - s_[:, 0]).mean()) / 100, ((ends.min(1).values - s_[:, 0]).mean()) / 100,
        ((ends.max(1).values - s_[:, 0]).mean()) / 100, sorted(set(((s_[:, 0] - t0) / 100).round().tolist()))[:6]))
