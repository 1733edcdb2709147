#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "PYTEST_EXIT=$rc"; tail -25 gpurun_out/pytest_gpu.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python bench_extra.py --which adw > gpurun_out/bench_extra_adw.log 2>&1
echo "EXTRA_EXIT=$?"; grep -v amdgpu.ids gpurun_out/bench_extra_adw.log | cut -c1-500
