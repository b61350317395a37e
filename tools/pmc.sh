#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc.sh <tag>  -> gpurun_out/pmc_<tag>/{fetch,write,sq}/...
# counters are collected in their own passes (no tracing domains besides kernel-trace), as the guide prescribes
set -e
tag=${1:-r01}
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-fast-mode > $out/$1.log 2>&1 || (tail -5 $out/$1.log; exit 1); }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS"
ls -R $out | head -40
