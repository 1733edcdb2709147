#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu | tail -25 | tee gpurun_out/gpu_all.txt
