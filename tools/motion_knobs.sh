#!/bin/bash
# A/B of the moving-camera feedback knobs: tools/motion_knobs.sh  (GPU box)
for wl in clouds_high_rm clouds_high; do
for knobs in "ATMO_FB_INSTREAM=1" "ATMO_FB_INSTREAM=0"; do
  echo "## $wl $knobs"
  env $knobs python tools/motion_sweep.py --workloads $wl --periods 8 --motions static,pan:0.1,pan:1,pan:5,orbit:0.1,orbit:1,orbit:5 2>&1 | grep -v "^#\|amdgpu.ids"
done
done
