#!/bin/bash
# In-kernel clock of the message kernels (MI355X guide, DVFS item 6): diagnostic build with (s_memtime, s_memrealtime) stamps around the
# block loop of every wave (variant libti_hip_stamps.so: tools/variant_bench.py build "stamps:-DTI_STAMPS+ONLY=..."), ~2 s of back-to-back
# launches on the bench data per arm; the clock is the median over the waves of layer 2's launch of the LAST evaluations.
#   usage: tools/gpu_clock.sh TAG
set -o pipefail
TAG=${1:-r03f}
OUT=gpurun_out/rec_$TAG
mkdir -p $OUT
LIB=$PWD/thermodynamic-interpolation_amd/build/variants/libti_hip_stamps.so
f=$OUT/${TAG}_inkernel_clock.txt
echo "# in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz around the row-block loop of every wave, layer-2 launch, headline batch (65536 x 18 atoms, F=128 L=5); median / p10 / p90 over waves; last 3 evaluations of a 12-step rollout" > $f
for arm in "f16x2 pair 65536" "f16x2 throughput 65536" "f32 pair 65536" "f32 throughput 65536" "f16 throughput 131072"; do
  set -- $arm
  echo "## precision $1, layout $2, batch $3" >> $f
  TI_LIB_PATH=$LIB TI_VB_TEMPLATE=$2 TI_STAMPS_DUMP=1 timeout -k 10 300 python3 tools/variant_bench.py child $3 12 $1 0 2>&1 >/dev/null | grep INKERNEL_CLOCK | tail -3 >> $f
done
cat $f
