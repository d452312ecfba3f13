#!/bin/bash
# round 6: the heavy-tile split's behaviour with the shipped 64 cost classes against 32 ('oc32'): its trigger reads the class histogram
cd /root/repo
export ROUNDS=3 STEPS=100
for wl in "clouds_high_rm P_space 1280 720" "clouds_high_rm P_limb" "clouds_high_rm P_night" "clouds_high_rm P_limb 1280 720" "clouds_high_rm P_clouds"; do
  tools/ab_bench.sh "$wl" base oc32
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
