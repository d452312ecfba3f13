#!/bin/bash
for c in "clouds_high_rm P_space 1920 1080" "clouds_high_rm P_space 3840 2160" "clouds_high_rm P_clouds 1920 1080" "direct32x8 P_space 1920 1080" "clouds_high P_space 1920 1080"; do
  ROUNDS=3 STEPS=60 tools/ab_bench.sh "$c" base stash stash7 stash7c4 stash8c4
done
