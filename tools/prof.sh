#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag>   -> gpurun_out/prof_<tag>/
set -e
tag=${1:-r01}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --no-fast-mode > $out/bench.log 2>&1
ls -R $out | head -30
