#!/bin/bash
# PMC passes for the bench (small batch), each in its own rocprofv3 run with kernel-trace only
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmc_list.txt 2>&1
CMD="python3 bench.py --batch 16384 --steps 1 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- $CMD > gpurun_out/pmc_$name.log 2>&1; echo "PMC_$name EXIT=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU || exit 1
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
run grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
find gpurun_out -name "*counter_collection.csv" | head
