#!/bin/bash
# kres.sh <file.hip> [extra hipcc flags] -- compile one translation unit for gfx950 and print per-kernel registers / spills / scratch
# (hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel).  Object goes to /tmp.
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -c "$src" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|VGPRs Spill|error" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - | sed -E 's/Function Name: //' | while read -r l; do
   n=$(echo "$l" | awk '{print $1}' | c++filt | sed -E 's/\(ti::EdgeParams\)//; s/void ti:://'); echo "$n | $(echo "$l" | cut -f2- | tr '\t' ' ')"; done
rm -f /tmp/kres_$$.o
