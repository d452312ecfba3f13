#!/bin/bash
# round 6, VERDICT r5 next #4: fewer gathers on today's kernels.  (i) ATMO_SHAPE_PAIR=1: planes k, k+1 of the shape volume in one 8-byte gather;
# (ii) ATMO_F4_SHAPE_MAX=64: the float copy of the 64^3 demo volume (two 16-byte gathers, no conversions).  Interleaved, 3 rounds, kernel ms.
O=gpurun_out/r6_gathers; mkdir -p $O
for spec in "clouds_high" "clouds_high_rm" "clouds_high P_space 3840 2160" "clouds_high_rm P_space 3840 2160" "clouds_high@lod0" "clouds_high_rm@lod0" "clouds_high_rm@lod0 P_space 3840 2160" "clouds_high P_limb" "clouds_high_rm P_ground"; do
  OFF=1 tools/ab_env.sh ATMO_SHAPE_PAIR "$spec" >> $O/ab_gathers.txt 2>&1
  OFF=64 tools/ab_env.sh ATMO_F4_SHAPE_MAX "$spec" >> $O/ab_gathers.txt 2>&1
done
cat $O/ab_gathers.txt
# the picture does not depend on either copy
ATMO_SHAPE_PAIR=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config or random_scenes or golden" > $O/parity_pair.txt 2>&1; tail -2 $O/parity_pair.txt
