#!/bin/bash
# Round-end check on the GPU box: the whole GPU suite, smoke(), and the DEFAULT bench command as the driver runs it (the final stdout line must
# be one parsable JSON record under 8 KB); copies of the line and the detail go to gpurun_out/final/ (commit them under profiles/round<N>/).
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/final/smoke.txt
T0=$(date +%s); ATMO_BENCH_DETAIL=gpurun_out/final/bench_default.json python bench.py > gpurun_out/final/bench_default_stdout.txt 2> gpurun_out/final/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
tail -1 gpurun_out/final/bench_default_stdout.txt > gpurun_out/final/bench_default_line.json
python - <<'PY'
import json
line = open("gpurun_out/final/bench_default_line.json").read()
d = json.loads(line)
print("final line:", len(line), "bytes; value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"], "valu", d.get("valu_roofline"), "cpu", d["cpu_baseline"]["value"])
print({k: v for k, v in d["extra"].items() if "moving" not in k})
PY
T0=$(date +%s); ATMO_BENCH_DETAIL= python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_cmd_stdout.txt 2>/dev/null; echo "driver-style bench rc=$? wall $(( $(date +%s) - T0 )) s, $(tail -1 gpurun_out/final/bench_driver_cmd_stdout.txt | wc -c) bytes"
