#!/bin/bash
O=gpurun_out/r6_step1; mkdir -p $O
python tests/checks/fuzz_four.py > $O/fuzz_four_after.txt 2>&1
tail -3 $O/fuzz_four_after.txt
python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; tail -5 $O/gpu_suite.txt
ATMO_FUZZ_EXTRA=1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_scenes > $O/fuzz_1212.txt 2>&1; tail -8 $O/fuzz_1212.txt
for wl in "clouds_high" "clouds_high_rm" "clouds_high_rm P_space 3840 2160" "clouds_high P_limb" "clouds_high_rm P_limb" "clouds_high@lod0" "clouds"; do
  tools/ab_bench.sh "$wl" base pre >> $O/ab_lambda_exact.txt 2>&1
done
cat $O/ab_lambda_exact.txt
