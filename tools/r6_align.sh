#!/bin/bash
# round 6: loop-header alignment (-falign-loops=32 / 64: 320 loop headers behind s_nop padding) and 64-byte alignment of the blocks nothing falls into
cd /root/repo
export ROUNDS=5 STEPS=100
for wl in "direct32x8" "shipped8" "lut32" "clouds_high" "clouds_high_rm" "clouds_high_rm P_space 3840 2160" "clouds_high P_limb" "clouds_high@lod0" "clouds_high_rm@lod0"; do
  tools/ab_bench.sh "$wl" base al32 al64 nft6
done
