#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc pass with SQ counters over bench.py (eager launches) into per-kernel MFMA utilisation.

  SQ_VALU_MFMA_BUSY_CYCLES = sum over the chip's 1024 SIMDs of the cycles their matrix pipe is busy (16 per v_mfma_f32_16x16x32_bf16,
  32 per 32x32x16: MI355X_MICROARCH.md, cycle constants); GRBM_GUI_ACTIVE / 8 = the dispatch's shader cycles (the counter sums the
  8 XCDs) - so  MFMA utilisation = MFMA_BUSY / (1024 * GRBM_GUI_ACTIVE / 8)  and  clock = GRBM_GUI_ACTIVE / 8 / duration."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
dur = {}
for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])):
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:84], r["Grid_Size"])
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[k]["_dur"].append(dur.get(r["Dispatch_Id"], 0.0))
rows = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    n = len(c.get("SQ_VALU_MFMA_BUSY_CYCLES", []))
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8
    if not n or cyc <= 0:
        continue
    # _dur holds one entry per counter row: the mean is the mean duration
    rows.append((n * m["_dur"], k, n, m, cyc))
rows.sort(reverse=True)
print("%-86s %8s %5s %9s %8s %9s %10s %9s %9s %8s" % ("kernel (eager, two branch streams)", "grid", "n", "dur_us", "GHz", "MFMA_util", "of_2.4GHz", "wait_any", "wait_inst", "lds_conf"))
for tot, k, n, m, cyc in rows[:14]:
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print("%-86s %8s %5d %9.1f %8.2f %8.1f%% %9.1f%% %8.1f%% %8.1f%% %8.0f" % (
        k[0], k[1], n, m["_dur"] * 1e6, cyc / m["_dur"] / 1e9 if m["_dur"] else 0.0, 100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
        100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["_dur"] * 2.4e9) if m["_dur"] else 0.0,
        100 * m.get("SQ_WAIT_ANY", 0.0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0.0) / wc, m.get("SQ_LDS_BANK_CONFLICT", 0.0)))
print("(MFMA_util = busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the GRBM quotient reads high on dispatches this short (GHz column above the 2.4 GHz maximum),"
      " so of_2.4GHz = busy / (1024 x duration x 2.4 GHz) is the fraction of the dense peak the kernel reaches while it shares the chip with the other branch)")
print("(wait_* = share of SQ_WAVE_CYCLES; kernels of the two graph branches run concurrently, so a kernel's duration includes the time it shares the chip)")
