#!/bin/bash
# round 6: what exact lambda + the non-power-of-two coordinate fix cost (base = this tree, pre = the commit before, vlog = exact arithmetic with v_log_f32,
# np2only = only the non-power-of-two fix), then the parity checks on the final form
O=gpurun_out/r6_step2; mkdir -p $O
for wl in "clouds_high" "clouds_high_rm" "clouds_high_rm P_space 3840 2160" "clouds_high P_limb" "clouds_high_rm P_limb" "clouds_high@lod0" "clouds_high_rm@lod0" "clouds"; do
  tools/ab_bench.sh "$wl" base pre vlog np2only >> $O/ab_lambda_exact.txt 2>&1
done
cat $O/ab_lambda_exact.txt
python tests/checks/fuzz_four.py > $O/fuzz_four_after.txt 2>&1; tail -1 $O/fuzz_four_after.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "log2_cr or exact_math or random_scenes or implicit" > $O/gpu_subset.txt 2>&1; tail -3 $O/gpu_subset.txt
