#!/bin/bash
# first GPU session: smoke -> parity tests -> small bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; echo "SMOKE_EXIT=$rc"; tail -5 gpurun_out/smoke.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "PYTEST_EXIT=$rc"; tail -40 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --batch 8192 --steps 2 --warmup 1 > gpurun_out/bench_small.log 2>&1
rc=$?; echo "BENCH_EXIT=$rc"; tail -5 gpurun_out/bench_small.log
