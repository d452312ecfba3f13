#!/bin/bash
# round 6's net effect: the final library ('base') against round 5's HEAD (86cb328: 'r5'), same box, interleaved, 5 rounds x 100 draws, kernel ms (bench.py, feedback on)
cd /root/repo
export ROUNDS=5 STEPS=100
for wl in "direct32x8" "direct32x8 P_space 3840 2160" "lut32" "shipped8" "clouds_high" "clouds_high P_space 3840 2160" "clouds_high_rm" "clouds_high_rm P_space 3840 2160" "clouds_high@lod0" "clouds_high_rm@lod0" "clouds_high_rm@lod0 P_space 3840 2160" "clouds_high P_limb" "clouds_high_rm P_limb" "clouds_high P_ground" "clouds_high_rm P_ground" "clouds_high_rm P_space 1280 720" "clouds"; do
  tools/ab_bench.sh "$wl" base r5
done
for m in orbit:1 pan:1; do for wl in direct32x8 clouds_high clouds_high_rm; do
  A=""; B=""
  for r in 1 2 3; do for v in base r5; do
    if [ $v = base ]; then unset ATMO_HIP_LIB; else export ATMO_HIP_LIB=$PWD/godot_atmosphere_shader_amd/libatmo_hip_$v.so; fi
    ms=$(ATMO_BENCH_DETAIL= python bench.py --workload $wl --motion $m --steps 128 --warmup 16 --no-cpu-baseline --also "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    if [ $v = base ]; then A="$A $ms"; else B="$B $ms"; fi
  done; done
  unset ATMO_HIP_LIB
  echo "$wl --motion $m (ms per step)   round 6:$A   round 5:$B"
done; done
