#!/bin/bash
# Builds an alternative libatmo_hip with extra hipcc flags for A/B runs:  tools/ab_build.sh <name> <flags...>
# then:  ATMO_HIP_LIB=$PWD/godot_atmosphere_shader_amd/libatmo_hip_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../godot_atmosphere_shader_amd/csrc"
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wall -Wno-unused-function "$@" \
  -o ../libatmo_hip_${name}.so atmo_api.hip atmo_kernels.hip
echo built ../libatmo_hip_${name}.so
