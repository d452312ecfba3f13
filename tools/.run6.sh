mkdir -p gpurun_out/r5f
(time python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "lane_split or heavy_tiles or tile_feedback or tile_list" 2>&1) > gpurun_out/r5f/pytest.log 2>&1
tail -12 gpurun_out/r5f/pytest.log
python tools/heavy_split_probe.py > gpurun_out/r5f/heavy_split_probe.txt 2>&1
cat gpurun_out/r5f/heavy_split_probe.txt
