#!/bin/bash
# round 6: the heavy-tile split's behaviour with the shipped 64 cost classes against 32 ('oc32'): its trigger reads the class histogram
cd /root/repo
export ROUNDS=3 STEPS=100
for wl in "clouds_high_rm P_space 1280 720" "clouds_high_rm P_limb" "clouds_high_rm P_night" "clouds_high_rm P_limb 1280 720" "clouds_high_rm P_clouds"; do
  tools/ab_bench.sh "$wl" base oc32
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
# one wave per workgroup (16 x 4 pixel tiles: the order's granularity is one wavefront) on the cloud kernels
python tests/checks/render_set.py /tmp/base.npz > /dev/null 2>&1; ATMO_HIP_LIB=$PWD/godot_atmosphere_shader_amd/libatmo_hip_th4.so python tests/checks/render_set.py /tmp/th4.npz > /dev/null 2>&1; python tests/checks/render_set.py --compare /tmp/base.npz /tmp/th4.npz | tail -2
export ROUNDS=3 STEPS=100
for wl in "clouds_high" "clouds_high_rm" "clouds_high_rm P_space 3840 2160" "clouds_high P_ground" "clouds_high_rm P_ground" "clouds_high@lod0" "clouds_high_rm@lod0" "clouds_high_rm P_space 1280 720" "direct32x8"; do
  tools/ab_bench.sh "$wl" base th4
done
