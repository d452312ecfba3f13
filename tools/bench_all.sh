#!/bin/bash
# Runs every workload at 1080p (and the two heavy ones at 4K) and prints one line per workload.
for wl in direct32x8 lut32 shipped8 clouds clouds_high clouds_high_rm; do
  python bench.py --workload $wl --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --also "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-16s %9.0f Mrays/s  kernel %.4f ms  hbm_frac %.4f' % ('$wl', d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac']))"
done
