#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r4c_pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r4c_pytest.txt
ATMO_FUZZ_EXTRA=240 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_scenes > gpurun_out/r4c_fuzz252.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r4c_fuzz252.txt
