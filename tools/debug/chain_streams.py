#!/usr/bin/env python3
"""How often do the two decoupled batch-slice chains fail to overlap?  Builds the config-2 loop N times and times 40 steps each, with the
loop's own stream being (a) the process's default stream, (b) a fresh pool stream.  A run near 4.4 ms/step = the chains ran back to back."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
c = bench.WORKLOADS[os.environ.get("WL", "c2")]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for policy in ("default", "pool", "default", "pool"):
    out = []
    for _ in range(n):
        if policy == "pool":
            with torch.cuda.stream(torch.cuda.Stream()):
                r = bench._time_loop(c, "bf16", dev, steps=40, warmup=5)
        else:
            r = bench._time_loop(c, "bf16", dev, steps=40, warmup=5)
        out.append(r["ms_per_step"])
    print("%-8s" % policy, " ".join("%.3f" % t for t in out), flush=True)
