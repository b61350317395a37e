#!/bin/bash
# PMC passes for the opt-in f16x3 mode (diagnostic): gpurun_out/pmc_<tag>/
set -e
tag=${1:-split}
export TMPDIR=/tmp
export FD_CONV_PRECISION=f16x3
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-fast-mode > $out/$1.log 2>&1 || (tail -5 $out/$1.log; exit 1); }
run sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
run sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"
run fetch "FETCH_SIZE"
