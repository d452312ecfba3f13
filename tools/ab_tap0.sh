#!/bin/bash
# Light tap 0 = the sample's own density (ATMO_RM_TAP0_REUSE): notap0 = -DATMO_RM_TAP0_REUSE=0
for c in "clouds_high_rm P_space 1920 1080" "clouds_high_rm P_space 3840 2160" "clouds_high_rm P_clouds 1920 1080" "clouds_high_rm_fast P_space 1920 1080"; do
  ROUNDS=3 STEPS=60 tools/ab_bench.sh "$c" notap0 base
done
