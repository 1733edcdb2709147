#!/bin/bash
# kernel trace + PMC passes of the bench for profiles/
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01b}
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/rocprof_$TAG.log 2>&1 || exit 1
tail -1 gpurun_out/rocprof_$TAG.log | cut -c1-300
CMD="python3 bench.py --batch ${PMC_BATCH:-16384} --steps 1 --warmup 1 --no-cpu-baseline $BENCH_ARGS"
run() { name=$1; shift; timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- $CMD > gpurun_out/pmc_${TAG}_$name.log 2>&1; echo "PMC_$name EXIT=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU || exit 1
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
run grbm GRBM_GUI_ACTIVE || exit 1
