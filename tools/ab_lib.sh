#!/bin/bash
# Same-box A/B of two library builds (how the round's kernel-level changes were judged: boxes of the pool differ by several percent,
# only alternating runs on ONE box compare).  Builds a second libmusehip.so from another revision and / or with extra compiler
# flags into tools/ab/ (git-ignored *.so; it travels to the GPU box with the tree), then on the box:
#     MUSEHIP_AB=1 MUSEHIP_LIB=$PWD/tools/ab/libmusehip_<tag>.so python bench.py --steps 100 --no-cpu-baseline --no-kernel-timing
# alternated with the plain command, three times each.
#   bash tools/ab_lib.sh head HEAD                      # the committed sources (against an edited working tree)
#   bash tools/ab_lib.sh nt0 . -DMH_PP_A_AUX=0          # the working tree with another cache policy for the full-row tile's A operand
# A revision older than an entry point that _lib.py binds needs a stub for it (the loader checks every declared symbol).
set -e
TAG=$1; REV=${2:-HEAD}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
mkdir -p $W/x/y "$ROOT/tools/ab"
cp -r "$ROOT/musediffusion_amd/csrc" $W/x/y/csrc
cp -r "$ROOT/include" $W/x/include
if [ "$REV" != "." ]; then
  for f in $(cd "$ROOT" && git ls-tree --name-only "$REV" musediffusion_amd/csrc/ | grep -E '\.(hip|h)$'); do
    (cd "$ROOT" && git show "$REV:$f") > $W/x/y/csrc/$(basename $f)
  done
  (cd "$ROOT" && git show "$REV:include/musehip.h") > $W/x/include/musehip.h
fi
cd $W/x/y/csrc && rm -f *.o libmusehip.so
make -j4 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*" > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
cp libmusehip.so "$ROOT/tools/ab/libmusehip_$TAG.so"
echo "$ROOT/tools/ab/libmusehip_$TAG.so"
