#!/bin/bash
# round 6 evidence run: rocprofv3 stats + PMC of every reported workload + the default bench (tools/profile_all.sh), the round-end check, the every-pixel 4K record
tools/profile_all.sh round6 > /dev/null 2>&1
ls gpurun_out/profiles_round6 | wc -l
bash tools/final_check.sh 2>&1 | tail -12
python tests/checks/every_pixel_4k.py 2>&1 | grep -v amdgpu.ids > gpurun_out/every_pixel_4k.txt; tail -1 gpurun_out/every_pixel_4k.txt
