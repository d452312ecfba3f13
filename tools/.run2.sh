mkdir -p gpurun_out/r5b
(time python -m pytest tests -m gpu -q -s 2>&1) > gpurun_out/r5b/pytest.log 2>&1
tail -5 gpurun_out/r5b/pytest.log
export ATMO_HIP_LIB_ABI=4
for wl in "direct32x8" "shipped8"; do ROUNDS=3 tools/ab_bench.sh "$wl" r4 base; done > gpurun_out/r5b/ab_after.txt 2>&1
cat gpurun_out/r5b/ab_after.txt
