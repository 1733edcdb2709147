#!/bin/bash
# rocprofv3 record of the exact-divergence workload (2048 molecules x 54 directions, F=128 L=5 A=18) for profiles/: kernel stats and
# separate --pmc passes (SQ, FETCH_SIZE, WRITE_SIZE, GRBM) on the torch-free tools/div_bench.py.   usage: tools/gpu_div_prof.sh TAG
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r03f}
OUT=gpurun_out/rec_$TAG
mkdir -p $OUT
timeout -k 10 300 python3 tools/div_bench.py 2048 3 f16x2 > $OUT/${TAG}_divergence_f16x2.json 2>/dev/null; echo "DIV_EXIT=$?"; cut -c1-300 $OUT/${TAG}_divergence_f16x2.json
timeout -k 10 300 python3 tools/div_bench.py 2048 3 f32 > $OUT/${TAG}_divergence_f32.json 2>/dev/null; echo "DIV32_EXIT=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_div -- python3 tools/div_bench.py 2048 2 f16x2 > $OUT/${TAG}_divergence_rocprof.log 2>&1 || { echo "ROCPROF_DIV FAILED"; exit 1; }
cp $(find gpurun_out/prof_${TAG}_div -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_divergence_kernel_stats.csv
run() { name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pq_${TAG}div_$name -- python3 tools/div_bench.py 2048 1 f16x2 > gpurun_out/pq_${TAG}div_$name.log 2>&1; echo "PMC_$name EXIT=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA || exit 1
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU || exit 1
run sq3 SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS SQ_WAVES SQ_INST_LEVEL_LDS || exit 1
run grbm GRBM_GUI_ACTIVE || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
python3 tools/pmc_quick.py ${TAG}div jvp > $OUT/${TAG}_divergence_pmc_summary.txt 2>&1
cat $OUT/${TAG}_divergence_pmc_summary.txt | head -40
