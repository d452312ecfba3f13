mkdir -p gpurun_out/r5h
python tools/heavy_split_probe.py > gpurun_out/r5h/heavy_split_probe_rm.txt 2>&1
cat gpurun_out/r5h/heavy_split_probe_rm.txt
