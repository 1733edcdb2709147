#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_divergence.py tests/test_data_drivers.py tests/test_host_logic.py -x -q 2>&1 | grep -v amdgpu | tail -15 | tee gpurun_out/div_tests.txt && \
timeout -k 10 600 python bench_extra.py --which div --steps 2 2>&1 | grep -v amdgpu | tee gpurun_out/bench_div.jsonl
