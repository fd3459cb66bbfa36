cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_train
mkdir -p $OUT
B="python3 bench.py --workload train --no-cpu-baseline --no-kernel-timing --steps 2 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -o p -- $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/p2 -o p -- $B > $OUT/p2.log 2>&1
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
tail -3 $OUT/p1.log $OUT/p2.log
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_train/p1", "gpurun_out/pmc_train/p2"):
    fs = glob.glob(d + "/*counter_collection.csv")
    if not fs: print("no counters in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "attn" in k or "gemm_tn" in k:
            print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, len(list(c.values())[0]))
PY
