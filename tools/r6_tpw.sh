#!/bin/bash
# round 6, VERDICT r5 next #6 candidate 2 in its cheapest form: the LUT no-cloud kernel shading 2 / 4 tiles per workgroup (half / a quarter of the waves,
# the per-wave preamble -- kernel-argument loads, tile index -- paid once per 2 / 4 tiles)
cd /root/repo
D=$PWD/godot_atmosphere_shader_amd
python tests/checks/render_set.py /tmp/base.npz > /dev/null 2>&1
for v in tpw2 tpw4; do ATMO_HIP_LIB=$D/libatmo_hip_$v.so python tests/checks/render_set.py /tmp/$v.npz > /dev/null 2>&1; python tests/checks/render_set.py --compare /tmp/base.npz /tmp/$v.npz | tail -2; done
export ROUNDS=5 STEPS=200
tools/ab_bench.sh "shipped8" base tpw2 tpw4
tools/ab_bench.sh "lut32" base tpw2 tpw4
tools/ab_bench.sh "shipped8 P_space 3840 2160" base tpw2 tpw4
tools/ab_bench.sh "shipped8 P_ground" base tpw2 tpw4
