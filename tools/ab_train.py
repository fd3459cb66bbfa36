#!/usr/bin/env python3
"""In-process A/B of the training tape options (training.FUSED_FFN) on bench.py --workload train: boxes of the pool differ by
several percent, so alternatives are only comparable inside one process."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--workload", "train", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-timing"]
import bench
from musediffusion_amd import training
import io, contextlib
for rnd in range(3):
    for fused in (True, False):
        training.FUSED_FFN = fused
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        import json
        d = json.loads(buf.getvalue().strip().splitlines()[-1])
        print("FUSED_FFN=%s: %.2f ms" % (fused, d["ms_per_step"]), flush=True)
