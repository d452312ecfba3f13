#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "geometric or feedback or moving_camera or graph or tile" 2>&1 | tail -2
export ROUNDS=5 STEPS=200
for wl in "direct32x8" "direct32x8 P_space 3840 2160" "lut32" "shipped8"; do tools/ab_bench.sh "$wl" base r5; done
for m in orbit:1 pan:1 pan:3; do A=""; B=""
  for r in 1 2 3; do for v in base r5; do
    if [ $v = base ]; then unset ATMO_HIP_LIB; else export ATMO_HIP_LIB=$PWD/godot_atmosphere_shader_amd/libatmo_hip_$v.so; fi
    ms=$(ATMO_BENCH_DETAIL= python bench.py --workload direct32x8 --motion $m --steps 128 --warmup 16 --no-cpu-baseline --also "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    if [ $v = base ]; then A="$A $ms"; else B="$B $ms"; fi
  done; done; unset ATMO_HIP_LIB
  echo "direct32x8 --motion $m (ms per step)   final:$A   round 5:$B"
done
