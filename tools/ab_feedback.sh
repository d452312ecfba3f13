#!/bin/bash
# Tile-order feedback off / on per workload, rate (whole step) and kernel time:  tools/ab_feedback.sh   (GPU box, repo root)
for c in "clouds_high_rm P_space 1920 1080" "clouds_high_rm P_space 3840 2160" "clouds_high_rm P_clouds 1920 1080" "clouds_high_rm P_ground 1920 1080" \
         "clouds_high P_space 1920 1080" "clouds_high P_clouds 1920 1080" "clouds_high P_ground 1920 1080" "clouds_high P_space 3840 2160" \
         "direct32x8 P_space 1920 1080" "direct32x8 P_space 3840 2160" "direct32x8 P_ground 1920 1080" "lut32 P_space 1920 1080" "shipped8 P_space 1920 1080"; do
  set -- $c
  for fb in 0 1 0 1; do
    r=$(ATMO_TILE_FEEDBACK=$fb python bench.py --workload $1 --pose $2 --width $3 --height $4 --steps 120 --warmup 16 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('kernel %.4f ms  rate %.0f Mrays/s' % (d['roofline']['kernel_avg_ms'], d['value']))")
    echo "$c feedback=$fb $r"
  done
done
