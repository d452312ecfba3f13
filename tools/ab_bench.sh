#!/bin/bash
# Interleaved A/B of alternative builds (tools/ab_build.sh):  tools/ab_bench.sh "<workload> [pose W H]" name1 name2 ...
# ("base" = the shipped libatmo_hip.so).  ROUNDS rounds (default 3), variants interleaved inside every round; prints
# the per-round kernel times (HIP events) and the median per variant.
set -u
read WL POSE W H <<< "$1"; shift
POSE=${POSE:-P_space}; W=${W:-1920}; H=${H:-1080}
D=$PWD/godot_atmosphere_shader_amd
declare -A T
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset ATMO_HIP_LIB; else export ATMO_HIP_LIB=$D/libatmo_hip_$v.so; fi
    name=${WL%@lod0}; samp=declared; [ "$name" != "$WL" ] && samp=lod0
    ms=$(ATMO_BENCH_EXPLICIT_SAMPLER=1 ATMO_BENCH_DETAIL= python bench.py --workload $name --sampler $samp --pose $POSE --width $W --height $H --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.readline())['roofline']['kernel_avg_ms'])")
    T[$v]="${T[$v]:-} $ms"
  done
done
unset ATMO_HIP_LIB
for v in "$@"; do
  echo "${T[$v]}" | python -c "
import sys; x=sorted(float(t) for t in sys.stdin.read().split()); print('%-28s %-10s %dx%d  %-14s median %.4f ms   rounds %s' % ('$WL', '$POSE', $W, $H, '$v', x[len(x)//2], ' '.join('%.4f'%t for t in x)))"
done
