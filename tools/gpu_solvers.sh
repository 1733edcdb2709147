#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_api.py tests/test_gpu_divergence.py -x -q 2>&1 | grep -v amdgpu | tail -30 | tee gpurun_out/solver_tests.txt
