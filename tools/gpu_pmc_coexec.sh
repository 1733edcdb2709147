#!/bin/bash
# one extra --pmc pass of the headline bench: how many cycles do matrix and vector instructions co-execute in the message kernel?
#   usage: tools/gpu_pmc_coexec.sh TAG   -> gpurun_out/rec_TAG/TAG_pmc_coexec.txt
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r03j}
OUT=gpurun_out/rec_$TAG
mkdir -p $OUT
CMD="python3 bench.py --batch 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-f32-leg"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_${TAG}_coexec -- $CMD > gpurun_out/pmc_${TAG}_coexec.log 2>&1; echo "PMC_coexec EXIT=$?"
python3 - "$TAG" > $OUT/${TAG}_pmc_coexec.txt <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/pmc_{tag}_coexec/**/*counter_collection.csv", recursive=True)[0]
D = collections.defaultdict(lambda: collections.defaultdict(float)); N = collections.Counter()
for r in csv.DictReader(open(f)):
    D[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
print("# per kernel, summed over its launches: fractions of SIMD time (4 x SQ_BUSY_CYCLES is not used: ratios of same-unit counters only)")
for k, m in D.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in m or m["SQ_VALU_MFMA_BUSY_CYCLES"] == 0: continue
    print(f"{k[:70]:70s} coexec / mfma_busy = {m['SQ_VALU_MFMA_COEXEC_CYCLES'] / m['SQ_VALU_MFMA_BUSY_CYCLES']:.3f}   active_valu / wave_cycles = {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.3f}"
          f"   active_lds / wave_cycles = {m['SQ_ACTIVE_INST_LDS'] / m['SQ_WAVE_CYCLES']:.3f}   vmem cycles / wave_cycles = {m['SQ_INST_CYCLES_VMEM'] / m['SQ_WAVE_CYCLES']:.3f}   transcendental insts = {m['SQ_INSTS_VALU_TRANS']:.3g}")
    print("     raw:", {c: v for c, v in m.items()})
PY
cat $OUT/${TAG}_pmc_coexec.txt | cut -c1-400
