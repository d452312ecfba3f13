#!/bin/bash
for c in "shipped8 P_space 1920 1080" "lut32 P_space 1920 1080" "clouds_high P_space 1920 1080" "shipped8 P_ground 1920 1080"; do
  ROUNDS=3 STEPS=100 tools/ab_bench.sh "$c" nodiv3 base
done
