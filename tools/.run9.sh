mkdir -p gpurun_out/r5i
(time python -m pytest tests -m gpu -q -s 2>&1) > gpurun_out/r5i/pytest.log 2>&1
tail -6 gpurun_out/r5i/pytest.log
PROBE_MODES=off,default,off,default python tools/heavy_split_probe.py 1280 720 > gpurun_out/r5i/probe_default.txt 2>&1
PROBE_MODES=off,default,off,default python tools/heavy_split_probe.py 1920 1080 >> gpurun_out/r5i/probe_default.txt 2>&1
cat gpurun_out/r5i/probe_default.txt
