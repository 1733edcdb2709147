#!/bin/bash
# Compile every kernel variant with resource remarks and list scratch (spill) bytes per lane; exits 1 if any kernel spills
# more than LIMIT bytes (default 1024; the F = 256 primal pass of the divergence sits at 0.8 KB and runs once per 75 tangent rows): a large number means hipcc lost the register allocation and the kernel will crawl.
LIMIT=${LIMIT:-1024}
cd "$(dirname "$0")/../thermodynamic-interpolation_amd/csrc" || exit 2
bad=0
for f in painn_kernels.hip painn_jvp_kernels.hip adw_kernels.hip ode_kernels.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    awk -v lim=$LIMIT -v file=$f '/Function Name:/ {name=$(NF-1)} /ScratchSize/ {n=$(NF-1); if (n+0 > 0) printf "%s %s scratch=%d B/lane\n", file, name, n; if (n+0 > lim) bad=1} END {exit bad}' || bad=1
done
exit $bad
