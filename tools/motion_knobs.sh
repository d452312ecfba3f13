#!/bin/bash
# A/B of the moving-camera feedback knobs on two motions: tools/motion_knobs.sh  (GPU box)
for wl in direct32x8 clouds_high_rm; do
for knobs in "ATMO_FB_MOVING_PERIOD=2 ATMO_FB_REACH_SCALE=1" "ATMO_FB_MOVING_PERIOD=8 ATMO_FB_REACH_SCALE=1" "ATMO_FB_MOVING_PERIOD=4 ATMO_FB_REACH_SCALE=1" \
             "ATMO_FB_MOVING_PERIOD=2 ATMO_FB_REACH_SCALE=2" "ATMO_FB_MOVING_PERIOD=2 ATMO_FB_REACH_SCALE=0.5" "ATMO_FB_MOVING_PERIOD=1 ATMO_FB_REACH_SCALE=1"; do
  echo "## $wl $knobs"
  env $knobs python tools/motion_sweep.py --workloads $wl --periods 8 --motions pan:0.1,pan:1,orbit:1 2>&1 | grep -v "^#"
done
done
