#!/bin/bash
# round 6: exact lambda, single pass, lean arithmetic (base = this tree, pre = the commit before, np2only = -DATMO_LOD_LAMBDA_EXACT=0, vlog = -DATMO_LOD_LOG2_CR=0)
O=gpurun_out/r6_step4; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "log2_cr or exact_math or random_scenes or implicit" > $O/gpu_subset.txt 2>&1; tail -3 $O/gpu_subset.txt
python tests/checks/fuzz_four.py > $O/fuzz_four_after.txt 2>&1; tail -1 $O/fuzz_four_after.txt
for wl in "clouds_high" "clouds_high_rm" "clouds_high_rm P_space 3840 2160" "clouds_high P_limb" "clouds_high_rm P_limb" "clouds_high@lod0" "clouds"; do
  tools/ab_bench.sh "$wl" base pre np2only vlog >> $O/ab_lambda_exact.txt 2>&1
done
cat $O/ab_lambda_exact.txt
