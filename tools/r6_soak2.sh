#!/bin/bash
# round 6: 10 000 FRESH random scenes (seeds 1212 .. 11211) through test_parity_random_scenes, default mode
ATMO_FUZZ_FIRST=1212 ATMO_FUZZ_EXTRA=10000 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_scenes --tb=line -p no:cacheprovider 2>&1 | grep -v amdgpu.ids > gpurun_out/fuzz_10000_fresh.txt
tail -15 gpurun_out/fuzz_10000_fresh.txt
