#!/bin/bash
# builds RingPipe variants on the GPU box and counts wrong molecules at a large batch (race hunt)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
cd thermodynamic-interpolation_amd/csrc
for v in "$@"; do
  flags=""; [ "$v" != "BASE" ] && flags=$(echo $v | sed 's/+/ -D/g; s/^/-D/')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTI_DEV_NB4_ONLY $flags -c painn_kernels.hip -o /tmp/pk.o 2>/dev/null || { echo "$v: compile failed"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libti_hip.so ../build/ti_api.o /tmp/pk.o ../build/adw_kernels.o || continue
  (cd ../.. && timeout -k 10 300 python tools/stress.py ${STRESS_B:-16384} ${STRESS_REPS:-4} $v 2>&1 | grep "bad molecules")
done
