#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python bench_extra.py "$@" > gpurun_out/bench_extra.log 2>&1
echo "EXTRA_EXIT=$?"; grep -v amdgpu.ids gpurun_out/bench_extra.log | cut -c1-400
