#!/bin/bash
# Interleaved A/B of one environment switch:  tools/ab_env.sh VAR "<workload> [pose W H]" [more workloads...]   (VAR=$OFF, default 0, against VAR unset)
VAR=$1; shift
for spec in "$@"; do
  read WL POSE W H <<< "$spec"; POSE=${POSE:-P_space}; W=${W:-1920}; H=${H:-1080}
  A=""; B=""
  for r in $(seq 1 ${ROUNDS:-3}); do
    for v in on off; do
      if [ $v = off ]; then export $VAR=${OFF:-0}; else unset $VAR; fi
      name=${WL%@lod0}; samp=declared; [ "$name" != "$WL" ] && samp=lod0
      ms=$(ATMO_BENCH_DETAIL= python bench.py --workload $name --sampler $samp --pose $POSE --width $W --height $H --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.readline())['roofline']['kernel_avg_ms'])")
      if [ $v = off ]; then B="$B $ms"; else A="$A $ms"; fi
    done
  done
  unset $VAR
  echo "$WL $POSE ${W}x$H   default:$A   $VAR=${OFF:-0}:$B"
done
