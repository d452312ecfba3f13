#!/bin/bash
# The round-3 A/B experiments whose records are under profiles/round3/ (GPU box, repo root):  tools/ab_round3.sh <experiment>
#   fast_miss   sure-miss test in front of the exact prologue, off / on per kernel family        -> ab_fast_miss.txt
#   div3        shared reciprocal for the three world divisions                                   -> ab_world_div3.txt
#   tap0        light tap 0 = the sample's own density                                           -> ab_rm_tap0.txt
#   taps        branch-free light taps                                                           -> ab_rm_tap0.txt
#   occupancy   lit-sample queue chunk 16 / 8 x launch bound none / 6 / 7                        -> ab_occupancy.txt
# Each builds the alternative libraries it needs (tools/ab_build.sh) and runs tools/ab_bench.sh: interleaved, 3 rounds, kernel ms.
set -u
B=tools/ab_build.sh
case "${1:-}" in
fast_miss)
  $B nofm -DATMO_FAST_MISS_MASK=0 >/dev/null; $B fmall -DATMO_FAST_MISS_MASK=0xff >/dev/null
  for c in "shipped8 P_space 1920 1080" "lut32 P_space 1920 1080" "direct32x8 P_space 1920 1080" "direct32x8 P_space 3840 2160" \
           "clouds_high P_space 1920 1080" "clouds_high_rm P_space 1920 1080" "v1_no_clouds P_space 1920 1080" "shipped8 P_ground 1920 1080"; do
    ROUNDS=3 STEPS=100 tools/ab_bench.sh "$c" nofm fmall base; done ;;
div3)
  $B nodiv3 -DATMO_WORLD_DIV3=0 >/dev/null
  for c in "shipped8 P_space 1920 1080" "lut32 P_space 1920 1080" "clouds_high P_space 1920 1080" "shipped8 P_ground 1920 1080"; do
    ROUNDS=3 STEPS=100 tools/ab_bench.sh "$c" nodiv3 base; done ;;
tap0)
  $B notap0 -DATMO_RM_TAP0_REUSE=0 >/dev/null
  for c in "clouds_high_rm P_space 1920 1080" "clouds_high_rm P_space 3840 2160" "clouds_high_rm P_clouds 1920 1080" "clouds_high_rm_fast P_space 1920 1080"; do
    ROUNDS=3 STEPS=60 tools/ab_bench.sh "$c" notap0 base; done ;;
taps)
  $B taps0 -DATMO_RM_TAPS_EARLY_OUT=0 >/dev/null
  for c in "clouds_high_rm P_space 1920 1080" "clouds_high_rm P_space 3840 2160" "clouds_high_rm P_clouds 1920 1080"; do
    ROUNDS=3 STEPS=60 tools/ab_bench.sh "$c" base taps0; done ;;
occupancy)
  $B c16 -DATMO_RMQ_CHUNK=16 -DATMO_MIN_WAVES=0 >/dev/null; $B c8 -DATMO_RMQ_CHUNK=8 -DATMO_MIN_WAVES=0 >/dev/null
  $B c16mw6 -DATMO_RMQ_CHUNK=16 -DATMO_MIN_WAVES=6 >/dev/null; $B c8mw7 -DATMO_RMQ_CHUNK=8 -DATMO_MIN_WAVES=7 >/dev/null
  for c in "clouds_high_rm P_space 1920 1080" "clouds_high_rm P_space 3840 2160" "clouds_high_rm P_clouds 1920 1080"; do
    ROUNDS=3 STEPS=60 tools/ab_bench.sh "$c" c16 c16mw6 c8 base c8mw7; done ;;
*) sed -n 2,9p "$0"; exit 1 ;;
esac
