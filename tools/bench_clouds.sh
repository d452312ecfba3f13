#!/bin/bash
for cfg in "clouds_high 1920 1080" "clouds_high_rm 1920 1080" "clouds_high_rm 3840 2160"; do set -- $cfg
  python bench.py --workload $1 --width $2 --height $3 --steps ${STEPS:-60} --warmup 6 --no-cpu-baseline --also "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-16s %sx%s %9.0f Mrays/s  kernel %.4f ms' % ('$1', '$2', '$3', d['value'], d['roofline']['kernel_avg_ms']))"
done
