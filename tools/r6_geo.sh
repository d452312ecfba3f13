#!/bin/bash
# round 6: the geometric tile order of the cloudless variants (ATMO_GEO_ORDER=1, default) against the learnt order (=0): static, orbit, pan; bench.py, 3 interleaved runs
cd /root/repo
for wl in direct32x8 lut32 shipped8; do for m in static orbit:1 pan:1 pan:3; do
  A=""; B=""
  for r in 1 2 3; do for v in 1 0; do
    if [ $m = static ]; then mo=""; else mo="--motion $m"; fi
    ms=$(ATMO_GEO_ORDER=$v ATMO_BENCH_DETAIL= python bench.py --workload $wl $mo --steps 128 --warmup 16 --no-cpu-baseline --also "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    if [ $v = 1 ]; then A="$A $ms"; else B="$B $ms"; fi
  done; done
  echo "$wl $m   geometric order:$A   learnt order:$B"
done; done
ATMO_GEO_ORDER=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "feedback or moving_camera or graph or stream or rect or tile" 2>&1 | tail -3
