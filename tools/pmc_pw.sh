#!/bin/bash
# HBM traffic of one conv shape (diagnostic): bash tools/pmc_pw.sh "128 512 1 1 80 80 1" [tile]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp FD_TILES=${2:-9} REPS=4
out=$PWD/gpurun_out/pmc_pw
rm -rf $out; mkdir -p $out
SHAPE="$1"
run() { timeout -k 10 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -o pmc -- python3 tools/time_conv.py $SHAPE > $out/$1.log 2>&1 || (tail -5 $out/$1.log; exit 1); }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python3 - <<PY
import csv, collections
for d in ("fetch", "write"):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open("$out/%s/pmc_counter_collection.csv" % d)):
        if "conv_igemm" not in r["Kernel_Name"]: continue
        e = rows.setdefault(r["Dispatch_Id"], {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    last = list(rows.values())[-1]
    print("$SHAPE", d, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in last.items()})
PY
