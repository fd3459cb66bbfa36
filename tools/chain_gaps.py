#!/usr/bin/env python3
"""Per-chain idle time inside a rocprofv3 --kernel-trace CSV of the decoupled sampler step: kernels are grouped by the queue they ran on, and for every
queue the gaps between one kernel's end and the next kernel's start are summed (tools/trace_gaps.py measures the UNION of all queues, which hides a chain's
own node-to-node latency behind the other chain's kernels).
    python tools/chain_gaps.py <kernel_trace.csv>
CAVEAT (round 5, measured): under rocprofv3 every hipGraph replay of a chain is followed by ~1.3 ms in which that queue runs nothing (24 such gaps = 32 % of a
chain's wall time in the trace).  That is the profiler, not the step: bench.py's own clock reads 3.50 ms per step with graph replays against 3.57 ms with eager
launches, whose per-kernel HIP-event spans add up to the wall time of both chains.  Inside a replay the node-to-node gaps are 0.1 - 0.3 us."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
byq = collections.defaultdict(list)
for r in rows:
    byq[r[qkey] if qkey else "0"].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]))
for q, ev in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    if len(ev) < 200:
        continue
    ev.sort()
    ev = ev[len(ev) // 4:]                      # skip warm-up / capture
    wall = ev[-1][1] - ev[0][0]
    busy = sum(e - s for s, e, _ in ev)
    gaps = [(ev[i + 1][0] - ev[i][1], ev[i][2], ev[i + 1][2]) for i in range(len(ev) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print("queue %s: %d kernels, wall %.2f ms, kernels %.2f ms, gaps %.2f ms (%.1f %%), median gap %.2f us, overlapping launches %d"
          % (q, len(ev), wall / 1e6, busy / 1e6, sum(g[0] for g in pos) / 1e6, 100.0 * sum(g[0] for g in pos) / wall,
             sorted(g[0] for g in pos)[len(pos) // 2] / 1e3 if pos else 0.0, len(gaps) - len(pos)))
    agg = collections.defaultdict(lambda: [0, 0])
    for g, a, b in pos:
        agg[(a, b)][0] += g; agg[(a, b)][1] += 1
    for (a, b), (ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:6]:
        print("    %8.1f us in %4d gaps (%.2f us each)  %s -> %s" % (ns / 1e3, n, ns / 1e3 / n, a, b))
