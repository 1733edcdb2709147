#!/bin/bash
# rehearsal of the N=2 launch line on ONE GPU (shared device, gloo): logic only, not a measurement
set -o pipefail
mkdir -p gpurun_out
TI_BENCH_TEST_SHARED_GPU=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --batch 4096 > gpurun_out/bench_mgpu_rehearsal.log 2>&1
echo "MGPU_EXIT=$?"; grep -v "amdgpu.ids\|^W\|^\*" gpurun_out/bench_mgpu_rehearsal.log | tail -5 | cut -c1-700
