#!/bin/bash
for wl in direct32x8 lut32 shipped8; do
  python bench.py --workload $wl --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --also "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-16s %9.0f Mrays/s  kernel %.4f ms' % ('$wl', d['value'], d['roofline']['kernel_avg_ms']))"
done
