#!/usr/bin/env python3
"""In-process A/B of a training tape option on bench.py --workload train (boxes of the pool differ by several percent, so
alternatives are only comparable inside one process):    python tools/ab_train.py TN_DW | FUSED_FFN | FUSED_ATTENTION"""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KNOB = sys.argv[1] if len(sys.argv) > 1 else "FUSED_FFN"
VALS = [int(v) for v in sys.argv[2:]]
sys.argv = ["bench.py", "--workload", "train", "--steps", os.environ.get("AB_STEPS", "4"), "--warmup", "2", "--no-cpu-baseline", "--no-kernel-timing"]
import bench  # noqa: E402
from musediffusion_amd import training  # noqa: E402

if KNOB in ("dw_wide", "gemm_variant", "auto_wide", "buf_dma", "strip"):      # library knobs: python tools/ab_train.py dw_wide 0 1 | gemm_variant 2 4 | strip 1 0
    from musediffusion_amd import _lib
    _lib.use_debug_library()   # the A/B switches live in libmusehip_dbg.so (include/musehip_dbg.h)
    vals = VALS
    for rnd in range(3):
        for v in vals:
            {"dw_wide": _lib.lib().mh_gemm_dw_set_wide, "gemm_variant": _lib.lib().mh_gemm_set_variant, "auto_wide": _lib.lib().mh_gemm_set_auto_wide, "buf_dma": _lib.lib().mh_gemm_set_buf_dma, "strip": _lib.lib().mh_gemm_set_strip}[KNOB](v)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                bench.main()
            print("%s=%d: %.2f ms" % (KNOB, v, json.loads(buf.getvalue().strip().splitlines()[-1])["ms_per_step"]), flush=True)
    _lib.lib().mh_gemm_dw_set_wide(1)
    _lib.lib().mh_gemm_set_variant(2)
    _lib.lib().mh_gemm_set_buf_dma(1)
    _lib.lib().mh_gemm_set_auto_wide(1)
    _lib.lib().mh_gemm_set_strip(1)
    raise SystemExit(0)
for rnd in range(3):
    for on in (True, False):
        setattr(training, KNOB, on)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        print("%s=%s: %.2f ms" % (KNOB, on, json.loads(buf.getvalue().strip().splitlines()[-1])["ms_per_step"]), flush=True)
setattr(training, KNOB, True)
