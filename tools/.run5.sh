mkdir -p gpurun_out/r5e
export ATMO_HIP_LIB_ABI=4
ATMO_HIP_LIB=$PWD/godot_atmosphere_shader_amd/libatmo_hip_r4.so python tests/checks/render_set.py /tmp/a.npz > gpurun_out/r5e/bits.log 2>&1
python tests/checks/render_set.py /tmp/b.npz >> gpurun_out/r5e/bits.log 2>&1
python tests/checks/render_set.py --compare /tmp/a.npz /tmp/b.npz >> gpurun_out/r5e/bits.log 2>&1
tail -8 gpurun_out/r5e/bits.log
(time python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "lane_split or heavy_tiles or tile_feedback or tile_list" 2>&1) > gpurun_out/r5e/pytest.log 2>&1
tail -25 gpurun_out/r5e/pytest.log
for wl in clouds_high_rm clouds_high; do for rep in 1 2 3; do for hs in 0 1; do
  echo -n "$wl ATMO_HEAVY_SPLIT=$hs: "
  ATMO_HEAVY_SPLIT=$hs ATMO_BENCH_DETAIL= python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done; done; done > gpurun_out/r5e/ab_heavy_split.txt 2>&1
cat gpurun_out/r5e/ab_heavy_split.txt
