#!/bin/bash
# the driver's command (K = 20, W = 5), events pre-created against created inside the region; 6 interleaved runs each
cd /root/repo
A=""; B=""
for r in 1 2 3 4 5 6; do for v in 0 1; do
  line=$(ATMO_BENCH_LAZY_EVENTS=$v ATMO_BENCH_DETAIL= python bench.py --steps 20 --warmup 5 --no-cpu-baseline --also "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.0f/%.4f/%.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']))")
  if [ $v = 0 ]; then A="$A $line"; else B="$B $line"; fi
done; done
echo "pre-created events (value / ms_per_step / kernel_avg_ms):$A"
echo "created inside the region:                              $B"
