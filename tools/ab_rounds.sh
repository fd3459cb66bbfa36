#!/bin/bash
# Same-box comparison of this tree with an earlier round's tree exported under tools/ab/<tag>/ (git archive <commit> | tar -x -C tools/ab/<tag>,
# then make in its csrc; tools/ab/ is git-ignored and travels with gpurun): alternates the two bench.py command lines, three rounds,
# and prints every ms_per_step.   bash tools/ab_rounds.sh r03 [extra bench flags]
TAG=${1:-r03}; shift
FLAGS="--steps 200 --warmup 10 --no-cpu-baseline --no-kernel-timing $*"
for r in 1 2 3; do
  (cd tools/ab/$TAG && python3 bench.py $FLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG   ', d['ms_per_step'], d['value'])")
  python3 bench.py $FLAGS --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('current', d['ms_per_step'], d['value'])"
done
