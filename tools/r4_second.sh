#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4b_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4b_pytest.txt
for spec in "direct32x8" "direct32x8 P_space 3840 2160" "direct32x8 P_ground"; do
  tools/ab_bench.sh "$spec" pre base cap02
done 2>&1 | tee gpurun_out/r4b_ab_sgpr.txt
for spec in "clouds_high@lod0" "clouds_high" "clouds_high_rm@lod0"; do
  tools/ab_bench.sh "$spec" base cap0a
done 2>&1 | tee -a gpurun_out/r4b_ab_sgpr.txt
