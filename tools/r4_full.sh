#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r4e_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4e_pytest.txt
for spec in "direct32x8" "lut32" "shipped8" "clouds_high@lod0" "clouds_high_rm@lod0 P_space 3840 2160"; do
  tools/ab_bench.sh "$spec" pre base
done 2>&1 | tee gpurun_out/r4e_ab.txt
