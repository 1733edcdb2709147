#!/bin/bash
# Quick SQ counter passes of the edge kernel on the variant-bench child (no torch): usage tools/pmc_quick.sh OUTTAG [LIBTAG] [BATCH]
# Separate --pmc passes with --kernel-trace only (no other trace domains).  Summary: tools/pmc_quick.py OUTTAG
set -o pipefail
export TMPDIR=/tmp
TAG=$1; LIBTAG=${2:-base}; B=${3:-16384}
mkdir -p gpurun_out
if [ "$LIBTAG" != "base" ]; then export TI_LIB_PATH=$PWD/thermodynamic-interpolation_amd/build/variants/libti_hip_$LIBTAG.so; fi
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pq_${TAG}_$name -- python3 tools/variant_bench.py child $B 1 ${PMC_PRECISION:-f16x2} 0 > gpurun_out/pq_${TAG}_$name.log 2>&1; echo "PMC_$name EXIT=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA || exit 1
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU || exit 1
run sq3 SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS SQ_WAVES SQ_INST_LEVEL_LDS || exit 1
run grbm GRBM_GUI_ACTIVE || exit 1
if [ -n "$PMC_DEEP" ]; then
  run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE || exit 1
  run if SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM || exit 1
  run lat SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL || exit 1
fi
if [ -n "$PMC_TRAFFIC" ]; then run fetch FETCH_SIZE || exit 1; run write WRITE_SIZE || exit 1; fi
python3 tools/pmc_quick.py $TAG
