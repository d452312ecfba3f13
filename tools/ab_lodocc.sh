#!/bin/bash
# declared-sampler raymarched-light kernel <51,0,1>: launch bound 4 (111 VGPRs) = lod4 against 5 (96 VGPRs, spills) = base
for sz in "1920 1080" "3840 2160"; do
 set -- $sz
 for r in 1 2 3; do for v in lod4 base; do
  if [ "$v" = base ]; then unset ATMO_HIP_LIB; else export ATMO_HIP_LIB=$PWD/godot_atmosphere_shader_amd/libatmo_hip_$v.so; fi
  ms=$(python bench.py --workload clouds_high_rm --sampler lod --width $1 --height $2 --steps 40 --warmup 8 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.readline())['roofline']['kernel_avg_ms'])")
  echo "clouds_high_rm@lod $1x$2 $v $ms"
 done; done
done
