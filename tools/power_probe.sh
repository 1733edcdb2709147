#!/bin/bash
# Samples board power and clocks (rocm-smi, sysfs only) while the torch-free bench child runs the headline batch: is the edge kernel
# running into the board's power limit (DVFS), i.e. is its time set by energy per row block rather than by issue slots or latency?
# usage (GPU box): tools/power_probe.sh TAG [PRECISION] [BATCH] [STEPS]
export TMPDIR=/tmp
TAG=$1; PREC=${2:-f16x2}; B=${3:-65536}; K=${4:-8}
mkdir -p gpurun_out
OUT=gpurun_out/power_${TAG}_${PREC}.txt
: > $OUT
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" > gpurun_out/power_${TAG}_idle.txt
timeout -k 10 300 python3 tools/variant_bench.py child $B $K $PREC 0 > gpurun_out/power_${TAG}_${PREC}_child.txt 2>&1 &
CH=$!
for i in $(seq 1 200); do
  kill -0 $CH 2>/dev/null || break
  echo "t=$(date +%s.%N)" >> $OUT
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" >> $OUT
  sleep 0.4
done
wait $CH
echo "child rc=$?"
grep -E "Power" $OUT | awk '{print $NF}' | sort -n | tail -3
