#!/bin/bash
# round 6: exact lambda in the march, rounds 2-5 arithmetic in the light taps (base = this tree, pre = the commit before round 6's kernels, np2only = -DATMO_LOD_LAMBDA_EXACT=0)
O=gpurun_out/r6_step7; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "log2_cr or exact_math or random_scenes or implicit or certificate" > $O/gpu_subset.txt 2>&1; tail -3 $O/gpu_subset.txt
python tests/checks/fuzz_four.py > $O/fuzz_four_after.txt 2>&1; tail -1 $O/fuzz_four_after.txt
for wl in "clouds_high" "clouds_high_rm" "clouds_high_rm P_space 3840 2160" "clouds_high P_limb" "clouds_high_rm P_limb" "clouds_high@lod0" "clouds"; do
  tools/ab_bench.sh "$wl" base pre np2only >> $O/ab_lambda_exact.txt 2>&1
done
cat $O/ab_lambda_exact.txt
ATMO_FUZZ_EXTRA=1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_scenes > $O/fuzz_1212.txt 2>&1; tail -4 $O/fuzz_1212.txt
python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt
python tests/checks/cert_soak.py 60 0 > $O/cert_soak.txt 2>&1; tail -2 $O/cert_soak.txt
