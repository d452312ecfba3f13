#!/bin/bash
# the headline kernel under a moving camera, tile-order feedback on / off (bench.py --motion, 128 poses, ms per step, 3 interleaved runs)
cd /root/repo
for wl in direct32x8 clouds_high; do for m in orbit:1 orbit:5 pan:1 pan:3; do
  A=""; B=""
  for r in 1 2 3; do for v in 1 0; do
    ms=$(ATMO_TILE_FEEDBACK=$v ATMO_BENCH_DETAIL= python bench.py --workload $wl --motion $m --steps 128 --warmup 16 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    if [ $v = 1 ]; then A="$A $ms"; else B="$B $ms"; fi
  done; done
  echo "$wl --motion $m   feedback on:$A   off:$B"
done; done
