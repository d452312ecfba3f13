#!/bin/bash
# __launch_bounds__(128, 6): the raymarched-cloud-light kernel at 80 instead of 84 VGPRs = 6 instead of 5 waves per SIMD
for c in "clouds_high_rm P_space 1920 1080" "clouds_high_rm P_space 3840 2160" "clouds_high_rm P_clouds 1920 1080" "clouds_high P_space 1920 1080" "direct32x8 P_space 1920 1080"; do
  ROUNDS=3 STEPS=60 tools/ab_bench.sh "$c" base mw6
done
