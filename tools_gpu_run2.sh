#!/bin/bash
# GPU session 2: full parity suite, full-size bench, rocprofv3 kernel trace of the bench
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "PYTEST_EXIT=$rc"; tail -15 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python bench.py > gpurun_out/bench_full.log 2>&1
rc=$?; echo "BENCH_EXIT=$rc"; tail -3 gpurun_out/bench_full.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/rocprof_bench.log 2>&1
rc=$?; echo "ROCPROF_EXIT=$rc"; tail -3 gpurun_out/rocprof_bench.log
find gpurun_out/prof_r01 -name "*stats*" | head; 
