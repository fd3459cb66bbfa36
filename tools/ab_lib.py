#!/usr/bin/env python3
"""A/B of two BUILDS of libmusehip.so on one box: alternates `bench.py` child processes that load the in-tree library and a saved
copy (MUSEHIP_AB=1 MUSEHIP_LIB=<copy>, the gate of musediffusion_amd/_lib.py) and prints the median ms/step of each.
    cp musediffusion_amd/csrc/libmusehip.so tools/ab/libmusehip_base.so     # before the change
    python tools/ab_lib.py tools/ab/libmusehip_base.so [--workload c2] [--rounds 3] [extra bench.py flags]"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("base")
ap.add_argument("--workload", default="c2")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=40)
a, extra = ap.parse_known_args()      # everything it does not know goes to bench.py (e.g. --dtype f16x3)
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "5", "--no-cpu-baseline", "--no-kernel-timing", "--no-secondary",
       "--workload", a.workload] + extra
res = {"base": [], "new": []}
for rnd in range(a.rounds):
    for which in ("base", "new"):
        env = dict(os.environ)
        if which == "base":
            env["MUSEHIP_AB"], env["MUSEHIP_LIB"] = "1", os.path.abspath(a.base)
        else:
            env.pop("MUSEHIP_AB", None)
            env.pop("MUSEHIP_LIB", None)
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
        if out.returncode:
            print(out.stderr[-2000:])
            raise SystemExit("bench.py failed with the %s library" % which)
        res[which].append(json.loads(out.stdout.strip().splitlines()[-1])["ms_per_step"])
        print("round %d %s: %.4f ms/step" % (rnd, which, res[which][-1]), flush=True)
for which in ("base", "new"):
    print("%s (%s): median %.4f ms/step (min %.4f)" % (which, a.workload, statistics.median(res[which]), min(res[which])))
print("new / base = %.4f" % (statistics.median(res["new"]) / statistics.median(res["base"])))
