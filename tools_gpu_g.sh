#!/bin/bash
for g in 1 2 3 5; do
  TI_FORCE_G=$g timeout -k 10 300 python bench.py --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G=$g', 'steps/s', round(r['value']), 'edge_ms', round(r['roofline']['avg_launch_ms'],2), 'upd_ms', round(r['roofline']['update_kernel_avg_ms'],2))"
done
