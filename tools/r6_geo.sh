#!/bin/bash
# round 6: the closed-form geometric tile order of the direct-light cloudless kernels where the learnt order has nothing (ATMO_GEO_ORDER=1, default) against =0
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "geometric or feedback or moving_camera or graph or stream or rect or tile" 2>&1 | tail -5
for wl in direct32x8; do for m in static orbit:1 orbit:5 pan:1 pan:3; do
  A=""; B=""
  for r in 1 2 3; do for v in 1 0; do
    if [ $m = static ]; then mo=""; else mo="--motion $m"; fi
    ms=$(ATMO_GEO_ORDER=$v ATMO_BENCH_DETAIL= python bench.py --workload $wl $mo --steps 128 --warmup 16 --no-cpu-baseline --also "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    if [ $v = 1 ]; then A="$A $ms"; else B="$B $ms"; fi
  done; done
  echo "$wl $m   geometric order:$A   learnt order only:$B"
done; done
for wl in "direct32x8 P_space 3840 2160" "direct32x8 P_limb" "shipped8" "lut32" "clouds_high"; do OFF=0 ROUNDS=3 STEPS=200 tools/ab_env.sh ATMO_GEO_ORDER "$wl"; done
