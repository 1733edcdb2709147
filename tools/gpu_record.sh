#!/bin/bash
# full record for profiles/: parity suite, default bench (with cpu baseline), rocprof kernel stats, PMC passes
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01f}
timeout -k 10 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "PYTEST_EXIT=$rc"; tail -3 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "SMOKE_EXIT=$?"; tail -1 gpurun_out/smoke.log
timeout -k 10 900 python bench.py > gpurun_out/bench_full.log 2>&1
rc=$?; echo "BENCH_EXIT=$rc"; tail -1 gpurun_out/bench_full.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
BENCH_ARGS="--no-f32-leg" tools/gpu_prof.sh $TAG
