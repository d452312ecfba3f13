#!/bin/bash
# Sure-miss test in front of the exact prologue, off (nofm = -DATMO_FAST_MISS_MASK=0) / on (base), per kernel family: tools/ab_fast_miss.sh
for c in "shipped8 P_space 1920 1080" "lut32 P_space 1920 1080" "direct32x8 P_space 1920 1080" "direct32x8 P_space 3840 2160" \
         "clouds_high P_space 1920 1080" "clouds_high_rm P_space 1920 1080" "v1_no_clouds P_space 1920 1080" "shipped8 P_ground 1920 1080" "direct32x8 P_ground 1920 1080"; do
  ROUNDS=3 STEPS=100 tools/ab_bench.sh "$c" nofm base
done
