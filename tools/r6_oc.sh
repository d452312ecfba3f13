#!/bin/bash
# round 6: 64 cost classes (shipped, 'base') against 32 ('oc32') under a moving camera: bench.py --motion, 128 poses, three interleaved runs each
cd /root/repo
for wl in clouds_high clouds_high_rm direct32x8; do for m in orbit:1 orbit:5 pan:1; do
  A=""; B=""
  for r in 1 2 3; do for v in base oc32; do
    if [ $v = base ]; then unset ATMO_HIP_LIB; else export ATMO_HIP_LIB=$PWD/godot_atmosphere_shader_amd/libatmo_hip_$v.so; fi
    ms=$(ATMO_BENCH_DETAIL= python bench.py --workload $wl --motion $m --steps 128 --warmup 16 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    if [ $v = base ]; then A="$A $ms"; else B="$B $ms"; fi
  done; done
  unset ATMO_HIP_LIB
  echo "$wl --motion $m   64 classes:$A   32 classes:$B"
done; done
