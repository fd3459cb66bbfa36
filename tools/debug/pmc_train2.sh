cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_train
mkdir -p $OUT
B="python3 bench.py --workload train --no-cpu-baseline --no-kernel-timing --steps 2 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/p3 -o p -- $B > $OUT/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_train/p3",):
    fs = glob.glob(d + "/*counter_collection.csv")
    if not fs: print("no counters in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "attn" in k or "gemm_tn" in k or "BigCfg<256, 256" in k:
            print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, len(list(c.values())[0]))
PY
