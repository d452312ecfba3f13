mkdir -p gpurun_out/r5g
python tools/heavy_split_probe.py > gpurun_out/r5g/heavy_split_probe.txt 2>&1
cat gpurun_out/r5g/heavy_split_probe.txt
