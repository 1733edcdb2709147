#!/bin/bash
# gpu_steps.sh "<cmd 1>" "<cmd 2>" ... -- run GPU steps one after another on a gpurun box.  A failing step (non-zero exit) does not
# stop the later ones, but a step that was killed by its timeout (exit 124 / 137) does: after a hung GPU step nothing else is started.
mkdir -p gpurun_out
i=0
for cmd in "$@"; do
    i=$((i + 1))
    echo "==== step $i: $cmd" | tee -a gpurun_out/steps.log
    bash -o pipefail -c "$cmd"
    rc=$?
    echo "==== step $i rc=$rc" | tee -a gpurun_out/steps.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "==== step $i timed out: stopping" | tee -a gpurun_out/steps.log
        exit $rc
    fi
done
exit 0
