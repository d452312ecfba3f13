#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "lod or LOD or declared or random_scenes or sampler or reference" > gpurun_out/r4d_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4d_pytest.txt
for spec in "clouds_high_rm" "clouds_high_rm P_space 3840 2160" "clouds_high_rm P_clouds" "clouds_high"; do
  tools/ab_bench.sh "$spec" pre base
done 2>&1 | tee gpurun_out/r4d_ab.txt
