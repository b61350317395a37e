#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_train.sh <tag>  -> gpurun_out/prof_<tag>/ : kernel stats of the training step
set -e
tag=${1:-train}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 tools/train_step_time.py > $out/run.log 2>&1
python3 - "$out/trace_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print(r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
