#!/bin/bash
# Builds libatmo_hip_<name>.so from the sources of another commit (the safe baseline for structural changes: a -DFLAG=0 build of the
# current source is only a baseline if it restores the old code generation).   tools/ab_build_commit.sh <name> <commit> [hipcc flags...]
# then:  gpurun -- 'tools/ab_bench.sh "<workload> [pose W H]" base <name>'
set -e
name=$1; commit=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/atmo_wt.XXXXXX)
git -C "$R" worktree add -f "$W" "$commit" -q
(cd "$W/godot_atmosphere_shader_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wall -Wno-unused-function "$@" \
   -o "$R/godot_atmosphere_shader_amd/libatmo_hip_${name}.so" atmo_api.hip atmo_kernels.hip)
git -C "$R" worktree remove --force "$W"
echo "built godot_atmosphere_shader_amd/libatmo_hip_${name}.so from $commit"
