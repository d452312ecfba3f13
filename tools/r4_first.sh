#!/bin/bash
# round 4, first GPU call: parity suite, the default bench line (length!), old-vs-new A/B of the groundwork commit
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4a_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4a_pytest.txt
T0=$(date +%s); ATMO_BENCH_DETAIL=gpurun_out/r4a_bench_detail.json python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench_line.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s, last line $(tail -1 gpurun_out/r4a_bench_line.json | wc -c) bytes, $(wc -l < gpurun_out/r4a_bench_line.json) lines"
tail -1 gpurun_out/r4a_bench_line.json | cut -c1-1500
for spec in "direct32x8" "direct32x8 P_space 3840 2160" "lut32" "shipped8" "clouds_high@lod0" "clouds_high_rm@lod0" "clouds_high_rm@lod0 P_space 3840 2160" "clouds_high" "clouds_high_rm P_space 3840 2160"; do
  tools/ab_bench.sh "$spec" pre base
done 2>&1 | tee gpurun_out/r4a_ab.txt
