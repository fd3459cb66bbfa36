#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) median/min duration and share."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    key = (n[:70], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
    d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values())
print("%-72s %7s %5s %5s %6s %6s %9s %9s %6s" % ("kernel", "blocks", "vgpr", "agpr", "lds", "n", "med_us", "min_us", "share"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    v = sorted(v)
    print("%-72s %7d %5s %5s %6s %6d %9.1f %9.1f %5.1f%%" % (k[0], k[1], k[2], k[3], k[4], len(v), v[len(v) // 2] / 1e3, v[0] / 1e3, 100 * sum(v) / tot))
