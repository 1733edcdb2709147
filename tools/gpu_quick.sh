#!/bin/bash
# parity suite for the drift kernels + headline bench (no CPU baseline, no f32 leg)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_divergence.py -x -q 2>&1 | grep -v amdgpu | tail -6 && \
timeout -k 10 600 python bench.py --no-cpu-baseline --no-f32-leg 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('steps/s', round(r['value']), 'edge_ms', round(r['roofline']['avg_launch_ms'],2), 'upd_ms', round(r['roofline']['update_kernel_avg_ms'],2), 'frac', round(r['roofline']['frac'],4))"
