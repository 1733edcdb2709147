#!/bin/bash
# secondary workloads (bench_extra.py) and a rocprofv3 kernel-stats record of the adw workload, for profiles/
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r03h}
OUT=gpurun_out/rec_$TAG
mkdir -p $OUT
timeout -k 10 900 python bench_extra.py > $OUT/${TAG}_bench_extra.jsonl 2>$OUT/${TAG}_bench_extra.err; echo "EXTRA_EXIT=$?"; cut -c1-260 $OUT/${TAG}_bench_extra.jsonl
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_adw -- python3 bench_extra.py --which adw --steps 5 > $OUT/${TAG}_adw_rocprof.log 2>&1 || { echo "ROCPROF_ADW FAILED"; exit 1; }
cp $(find gpurun_out/prof_${TAG}_adw -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_adw_kernel_stats.csv
head -5 $OUT/${TAG}_adw_kernel_stats.csv | cut -c1-200
