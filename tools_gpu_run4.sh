#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "PYTEST_EXIT=$rc"; grep "split-fp16" gpurun_out/pytest_gpu.log | head -40; tail -5 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench_full.log 2>&1
rc=$?; echo "BENCH_EXIT=$rc"; tail -2 gpurun_out/bench_full.log | cut -c1-1600
