#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py into per-kernel HBM bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o p -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o p -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w c2/bf16 > gpurun_out/pmc_traffic_summary.txt

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts wide streaming reads at half their bytes
(MI355X_MICROARCH.md, HBM), so it is doubled.  Writes gpurun_out/pmc_traffic.json for bench.py's `roofline.traffic`.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""), r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


import re


def variant_of(kernel_name):
    """rocprof kernel name -> the variant string bench.py reports for its dominant kernel (tile + epilogue signature)"""
    m = re.search(r"gemm_big_kernel<BigCfg<(\d+), (\d+), \d+, \d+, \d+, (true|false)>, (\d+), (\d+), (\d+)>", kernel_name)
    if not m:
        return kernel_name
    bm, bn, pp, epi, act, dbg = m.groups()
    role = {"3": "dense+bias+residual+LayerNorm", "1": "QKV projection + head scatter"}.get(epi) or (
        "dense+GELU (FFN1)" if act == "2" else "dense+tanh" if act == "1" else "dense")
    return "gemm_big_kernel<%sx%s%s, EPI=%s> bf16 %s" % (bm, bn, "pp" if pp == "true" else "", epi, role)


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
key = sys.argv[3]
outdir = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out"
rows = []
for k in fetch:
    fk, n = fetch[k]
    wk = write.get(k, (0.0, 0))[0]
    rows.append((n * (2 * fk + wk), k, n, fk, wk))
rows.sort(reverse=True)
print("%-78s %8s %6s %12s %12s %14s" % ("kernel", "grid", "n", "FETCH KiB", "WRITE KiB", "HBM MB/launch"))
for tot, k, n, fk, wk in rows[:14]:
    print("%-78s %8s %6d %12.1f %12.1f %14.2f" % (k[0][:78], k[1], n, fk, wk, (2 * fk + wk) * 1024 / 1e6))
# (the last template argument carries build bits - 16384 = stage DMA as buffer loads - that do not change what the kernel is)
dom = [r for r in rows if re.search(r"gemm_big_kernel<BigCfg<[^>]*>, 3, 0, (0|16384)>", r[1][0])]
if len(sys.argv) > 5 and sys.argv[5] == "pool":    # the training step: bench.py's dominant symbol is the pooled 256x128 EPI 0 launch, not the LayerNorm tile
    dom = []
if not dom:
    # no full-row LayerNorm tile at this width (d_model 768): bench.py's dominant symbol pools every EPI 0 / no-activation launch of the
    # 256x128 tile, whatever its deferred-LayerNorm variant (last template argument) and shape - pool them the same way
    pool = [r for r in rows if re.search(r"gemm_big_kernel<BigCfg<256, 128, 2, 2, 3, false>, 0, 0, \d+>", r[1][0])]
    if pool:
        n = sum(r[2] for r in pool)
        fk = sum(r[2] * r[3] for r in pool) / n
        wk = sum(r[2] * r[4] for r in pool) / n
        dom = [(0, (pool[0][1][0], "pooled over %d variants / shapes" % len(pool)), n, fk, wk)]
if dom:
    tot, k, n, fk, wk = dom[0]
    rec = {key: {"kernel": k[0], "variant": variant_of(k[0]), "launches_sampled": n, "fetch_kib": fk, "write_kib": wk,
                 "traffic_bytes_per_launch": (2 * fk + wk) * 1024,
                 "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950)"}}
    json.dump(rec, open(outdir + "/pmc_traffic.json", "w"), indent=1)
    print("dominant kernel (%s): %.1f MB per launch" % (k[0][:70], rec[key]["traffic_bytes_per_launch"] / 1e6))
