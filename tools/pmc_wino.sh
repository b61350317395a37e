#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_wino.sh <tag> [shape] [2|4] -> gpurun_out/pmc_<tag>/{sq,lds,fetch,write}: PMC passes of the Winograd kernel
set -e
tag=${1:-wino}; shape=${2:-tower}; alg=${3:-2}
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -o pmc -- python3 tools/wino_one.py $shape $alg > $out/$1.log 2>&1 || (tail -5 $out/$1.log; exit 1); }
run sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python3 - <<PY
import csv, collections
for d in ("sq", "lds", "fetch", "write"):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open("$out/%s/pmc_counter_collection.csv" % d)):
        if "wino_kernel" not in r["Kernel_Name"] and "wino4_kernel" not in r["Kernel_Name"]: continue
        e = rows.setdefault(r["Dispatch_Id"], {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    last = list(rows.values())[-1]
    print(d, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in last.items()})
PY
