#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_train.sh <tag>  -> gpurun_out/prof_<tag>/ : kernel trace of tools/train_step_time.py
# + last_step_kernel_stats.csv (per-kernel summary of the final training step)
set -e
tag=${1:-train}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 tools/train_step_time.py > $out/run.log 2>&1
grep "train step" $out/run.log
ms=$(grep "train step" $out/run.log | sed 's/.*: \([0-9.]*\) ms.*/\1/')
python3 tools/trace_last_step.py $out/trace_kernel_trace.csv $ms $out/last_step_kernel_stats.csv
