#!/bin/bash
# builds variants of the tangent kernels on the GPU box and counts outlier molecules (race hunt)
set -o pipefail
export TMPDIR=/tmp
cd thermodynamic-interpolation_amd/csrc
for v in "$@"; do
  flags=""; [ "$v" != "BASE" ] && flags=$(echo $v | sed 's/+/ -D/g; s/^/-D/')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTI_DEV_NB4_ONLY $flags -c painn_jvp_kernels.hip -o /tmp/jk.o 2>/dev/null || { echo "$v: compile failed"; continue; }
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTI_DEV_NB4_ONLY $flags -c painn_kernels.hip -o /tmp/pk.o 2>/dev/null || { echo "$v: compile failed"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libti_hip.so ../build/ti_api.o /tmp/pk.o /tmp/jk.o ../build/adw_kernels.o ../build/ode_kernels.o || continue
  (cd ../.. && TI_TEMPLATE=throughput timeout -k 10 400 python tools/stress_div.py ${STRESS_B:-256} ${STRESS_REPS:-6} $v 2>&1 | grep "outlier")
done
