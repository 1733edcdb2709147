#!/bin/bash
# Full record for profiles/ (one gpurun call): default bench line, rocprofv3 kernel stats of the same command for both matrix
# paths, separate --pmc passes at the headline batch (SQ, FETCH_SIZE, WRITE_SIZE, GRBM) -> per-kernel table + traffic JSON.
#   gpurun --timeout 1200 -- tools/gpu_record.sh r02a     then copy gpurun_out/rec_r02a/* into profiles/
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r02a}
OUT=gpurun_out/rec_$TAG
mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/${TAG}_bench_full.json.log 2>&1; rc=$?; echo "BENCH_EXIT=$rc"; tail -1 $OUT/${TAG}_bench_full.json.log | cut -c1-400
[ $rc -ne 0 ] && exit $rc
for prec in f16x2 f32; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$prec -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-f32-leg --precision $prec > $OUT/${TAG}_rocprof_$prec.log 2>&1 || { echo "ROCPROF_$prec FAILED"; exit 1; }
  cp $(find gpurun_out/prof_${TAG}_$prec -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats_$prec.csv
  tail -1 $OUT/${TAG}_rocprof_$prec.log | cut -c1-200
done
export PMC_BATCH=65536 PMC_PRECISION=f16x2
CMD="python3 bench.py --batch $PMC_BATCH --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-f32-leg"
run() { name=$1; shift; timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- $CMD > gpurun_out/pmc_${TAG}_$name.log 2>&1; echo "PMC_$name EXIT=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU || exit 1
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
run grbm GRBM_GUI_ACTIVE || exit 1
python3 tools/pmc_summary.py $TAG $OUT/${TAG}_pmc_summary.md
cp profiles/pmc_edge_traffic.json $OUT/pmc_edge_traffic.json
