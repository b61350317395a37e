#!/bin/bash
# PMC passes over tools/time_wgrad.py (diagnostic): gpurun_out/pmc_wgrad_<tag>/
set -e
tag=${1:-a}
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_wgrad_$tag
mkdir -p $out
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -o pmc -- python3 tools/time_wgrad.py 0 > $out/$1.log 2>&1 || (tail -5 $out/$1.log; exit 1); }
run sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
run sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"
python3 - $out <<'PY'
import csv, sys, collections
out = sys.argv[1]
for p in ("sq", "sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f"{out}/{p}/pmc_counter_collection.csv")):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in agg.items():
        if "wgrad" in k:
            print(p, k, {a: f"{b:.3g}" for a, b in v.items()})
PY
