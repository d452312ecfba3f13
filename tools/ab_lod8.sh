#!/bin/bash
for r in 1 2 3; do for v in base lod8; do
  if [ "$v" = base ]; then unset ATMO_HIP_LIB; else export ATMO_HIP_LIB=$PWD/godot_atmosphere_shader_amd/libatmo_hip_$v.so; fi
  ms=$(python bench.py --workload clouds_high --sampler lod --steps 60 --warmup 8 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.readline())['roofline']['kernel_avg_ms'])")
  echo "clouds_high@lod 1920x1080 $v $ms"
done; done
