#!/usr/bin/env python3
"""In-process A/B of library switches on the captured reverse step (bench.py workload c2): alternates settings over several
rounds and prints the median ms/step of each (boxes of the pool differ by several percent; only same-process numbers compare).
    python tools/ab_step.py plain_stores 0 1 2 4 7          (AB_WORKLOAD=c2-bertbase selects another bench workload; AB_ROUNDS / AB_STEPS: alternations and timed steps, default 3 x 40)"""
import contextlib
import io
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
knob, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
BASE = ["bench.py", "--steps", os.environ.get("AB_STEPS", "40"), "--warmup", "5", "--no-cpu-baseline", "--no-kernel-timing", "--no-secondary", "--workload", os.environ.get("AB_WORKLOAD", "c2")]
sys.argv = list(BASE)
import bench  # noqa: E402
from musediffusion_amd import _lib  # noqa: E402
_lib.use_debug_library()   # the A/B switches live in libmusehip_dbg.so (include/musehip_dbg.h)

setters = {"gemm_debug": lambda v: _lib.lib().mh_gemm_set_debug(v), "plain_stores": lambda v: _lib.lib().mh_gemm_set_plain_stores(v), "fuse_ln": lambda v: _lib.lib().mh_denoiser_set_fuse_ln(v),
           "v4_split": None, "skip": lambda v: _lib.lib().mh_denoiser_set_skip(v), "ln_rows4": lambda v: _lib.lib().mh_layernorm_set_rows4(v), "prescale_q": lambda v: _lib.lib().mh_denoiser_set_prescale_q(v),
           "stream_attn": lambda v: _lib.lib().mh_attention_set_stream(v), "wide_roles": lambda v: _lib.lib().mh_gemm_set_wide_roles(v), "fuse_headtail": lambda v: _lib.lib().mh_denoiser_set_fuse_headtail(v), "gemm_variant": lambda v: _lib.lib().mh_gemm_set_variant(v), "buf_dma": lambda v: _lib.lib().mh_gemm_set_buf_dma(v)}
from musediffusion_amd.models.diffusion import GaussianDiffusion  # noqa: E402
setters["strip"] = lambda v: _lib.lib().mh_gemm_set_strip(v)               # 1 FFN1 on the column-strip kernel (round 6) / 0 on the 256 x 128 tile
setters["defer_ln"] = lambda v: _lib.lib().mh_denoiser_set_defer_ln(v)     # 0 never / 1 widths without a full-row tile / 2 always
setters["decouple"] = lambda v: setattr(GaussianDiffusion, "decouple_branches", bool(v))
setters["shared"] = lambda v: setattr(GaussianDiffusion, "shared_head_tail", bool(v))
setters["fuse_noise"] = lambda v: setattr(GaussianDiffusion, "fuse_noise", bool(v))
setters["update_in_forward"] = lambda v: setattr(GaussianDiffusion, "update_in_forward", bool(v))
setters["round_in_forward"] = lambda v: setattr(GaussianDiffusion, "round_in_forward", bool(v))
setters["fuse_rounding"] = lambda v: setattr(GaussianDiffusion, "fuse_rounding", bool(v))
setters["skew"] = lambda v: setattr(GaussianDiffusion, "branch_skew_us", None if v < 0 else v)       # microseconds; -1 = automatic
res = {v: [] for v in values}
for rnd in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for v in values:
        if knob == "v4_split":      # value = 10 * variant + branches: 21 = default tiles, one branch; 41 = 256x256 tile, one branch; 42 ...
            _lib.lib().mh_gemm_set_variant(v // 10)
            sys.argv = BASE + ["--split", str(v % 10)]
        else:
            setters[knob](v)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        res[v].append(json.loads(buf.getvalue().strip().splitlines()[-1])["ms_per_step"])
for v in values:
    print("%s=%d: median %.4f ms/step (min %.4f)%s" % (knob, v, statistics.median(res[v]), min(res[v]),
                                                       "  all: " + " ".join("%.3f" % t for t in res[v]) if os.environ.get("AB_ALL") else ""), flush=True)
