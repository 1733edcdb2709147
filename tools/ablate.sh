#!/bin/bash
# builds ablation variants of libti_hip.so on the GPU box (hipcc is there) and times the small bench with each
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
cd thermodynamic-interpolation_amd/csrc
for v in BASE "$@"; do
  flags=""; [ "$v" != "BASE" ] && flags=$(echo $v | sed 's/+/ -D/g; s/^/-D/')
  case "$v" in FLAG:*) flags="${v#FLAG:}";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTI_DEV_NB4_ONLY $flags -c painn_kernels.hip -o /tmp/pk.o 2>/dev/null || { echo "$v: compile failed"; continue; }
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c ti_api.hip -o /tmp/api.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libti_hip.so /tmp/api.o /tmp/pk.o ../build/painn_jvp_kernels.o ../build/adw_kernels.o ../build/ode_kernels.o || continue
  (cd ../.. && TI_IGNORE_NAN=1 TI_BENCH_NOCHECK=1 timeout -k 10 300 python bench.py --batch 32768 --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'steps/s', round(r['value']), 'edge_ms', round(r['roofline']['avg_launch_ms'],2), 'frac', round(r['roofline']['frac'],3), 'upd_ms', round(r['roofline']['update_kernel_avg_ms'],2))")
done
