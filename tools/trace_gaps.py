#!/usr/bin/env python3
"""Idle time of the GPU inside a rocprofv3 --kernel-trace CSV: the union of all kernel intervals against the wall span, and where the
gaps are (which kernel ended last before a gap, which started after it).
    python tools/trace_gaps.py <kernel_trace.csv> [first_kernel_substring]
With the second argument only the region from the first launch whose name contains it (e.g. step_begin) is analysed."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60])
             for r in rows), key=lambda e: e[0])
if len(sys.argv) > 2:
    idx = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
    ev = ev[idx[len(idx) // 4]: idx[-1]]          # skip warm-up: from the first quarter's marker to the last marker
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy, cur_end, gaps = 0, ev[0][0], collections.Counter()
gap_ns = collections.Counter()
last_name = ev[0][2]
conc = 0
for s, e, n in ev:
    if s > cur_end:
        gaps[(last_name, n)] += 1
        gap_ns[(last_name, n)] += s - cur_end
        busy += 0
        cur_end = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        last_name = n
wall = t1 - t0
print("wall %.3f ms, busy (union of kernel intervals) %.3f ms = %.2f %%, idle %.3f ms; sum of kernel durations %.3f ms (x%.2f overlap)"
      % (wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, sum(e - s for s, e, _ in ev) / 1e6, sum(e - s for s, e, _ in ev) / busy))
print("largest idle gaps by (kernel that ended last -> kernel that started next):")
for k, ns in gap_ns.most_common(12):
    print("  %8.1f us in %4d gaps (%.2f us each)  %s  ->  %s" % (ns / 1e3, gaps[k], ns / 1e3 / gaps[k], k[0], k[1]))
