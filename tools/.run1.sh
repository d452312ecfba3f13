set -x
mkdir -p gpurun_out/r5a
(time python -m pytest tests -m gpu -x -q -s 2>&1) > gpurun_out/r5a/pytest.log 2>&1
tail -5 gpurun_out/r5a/pytest.log
export ATMO_HIP_LIB_ABI=4
for wl in "direct32x8" "shipped8" "lut32" "clouds_high" "clouds_high_rm"; do ROUNDS=3 tools/ab_bench.sh "$wl" r4 base; done > gpurun_out/r5a/ab_tile_bound.txt 2>&1
cat gpurun_out/r5a/ab_tile_bound.txt
