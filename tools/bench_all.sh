#!/bin/bash
# One line per workload: tools/bench_all.sh [pose] [width height]   (STEPS=100 by default)
POSE=${1:-P_space}; W=${2:-1920}; H=${3:-1080}
for wl in ${WORKLOADS:-direct32x8 lut32 shipped8 clouds_high clouds_high_rm clouds_high@lod0 clouds_high_rm@lod0 clouds_high_fast clouds_high_rm_fast}; do
  name=${wl%@lod0}; samp=declared; [ "$name" != "$wl" ] && samp=lod0
  ATMO_BENCH_DETAIL= python bench.py --workload $name --sampler $samp --pose $POSE --width $W --height $H --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --also "" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-20s %-8s %dx%d %9.0f Mrays/s  kernel %.4f ms  %s' % ('$wl', '$POSE', $W, $H, d['value'], d['roofline']['kernel_avg_ms'], d['config']['kernel']))"
done
