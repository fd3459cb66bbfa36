#!/bin/bash
# Round profile evidence, run on the GPU box from the repo root:  bash tools/profile_round.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (hipGraph replay);
# 2. separate --pmc passes (eager launches, --no-graph): FETCH_SIZE | WRITE_SIZE | SQ MFMA / busy / wait counters + GRBM clock;
# 3. text summaries under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
# (rocprofv3 runs python3 directly after `--`: no env / bash -c hop; counters never together with sys/hip traces)
TAG=${1:-r02}
WL=${2:-c2}
OUT=gpurun_out/${TAG}_${WL}
mkdir -p $OUT
B="python3 bench.py --workload $WL --no-cpu-baseline --no-kernel-timing --no-secondary"
STEPS=30; [ "$WL" = "train" ] && STEPS=10
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- $B --steps $STEPS --warmup 3 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -o p -- $B --steps 3 --warmup 1 --no-graph > $OUT/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -o p -- $B --steps 3 --warmup 1 --no-graph > $OUT/pmc_w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $B --steps 3 --warmup 1 --no-graph > $OUT/pmc_sq.log 2>&1
python3 tools/kstats.py $(ls $OUT/trace/*kernel_trace.csv | head -1) $STEPS > $OUT/kernel_trace_summary.txt 2>&1
cp $(ls $OUT/trace/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null
if [ "$WL" = "train" ]; then
  python3 tools/pmc_traffic.py $OUT/pmc_f $OUT/pmc_w train/bf16/b1 $OUT pool > $OUT/pmc_traffic_summary.txt 2>&1
else
  python3 tools/pmc_traffic.py $OUT/pmc_f $OUT/pmc_w $WL/bf16/b2 $OUT > $OUT/pmc_traffic_summary.txt 2>&1
fi
python3 tools/pmc_sq.py $OUT/pmc_sq > $OUT/pmc_mfma_summary.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_f $OUT/pmc_w $OUT/pmc_sq   # (raw traces: hundreds of MB; the summaries above are what is kept)
ls $OUT
