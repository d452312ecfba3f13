#!/bin/bash
# round 6: 3 000 fresh random scenes (seeds 1212 .. 4211) in the reference-order mode (atmo_set_precision 2; bars 1e-5 cloudless / 5e-5 cloud variants)
ATMO_FUZZ_PRECISE=1 ATMO_FUZZ_FIRST=1212 ATMO_FUZZ_EXTRA=3000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k random_scenes --tb=line -p no:cacheprovider 2>&1 | grep -v amdgpu.ids > gpurun_out/fuzz_3000_precise.txt
tail -12 gpurun_out/fuzz_3000_precise.txt | cut -c1-400
