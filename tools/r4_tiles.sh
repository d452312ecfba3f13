#!/bin/bash
mkdir -p gpurun_out
python tools/band_balance.py lod0 split2 > gpurun_out/r4f_band_balance_lod0_split2.txt 2>&1; tail -8 gpurun_out/r4f_band_balance_lod0_split2.txt
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
ATMO_BENCH_FORCE_DIST=1 ATMO_BENCH_DETAIL= python bench.py --steps 20 --warmup 5 --shard tiles --workload clouds_high_rm > gpurun_out/r4f_tiles1.out 2> gpurun_out/r4f_tiles1.err; echo "tiles rc=$?"; tail -1 gpurun_out/r4f_tiles1.out | cut -c1-300
ATMO_BENCH_FORCE_DIST=1 ATMO_BENCH_DETAIL= python bench.py --steps 20 --warmup 5 > gpurun_out/r4f_dist1.out 2> gpurun_out/r4f_dist1.err; echo "dist1 rc=$?"; tail -1 gpurun_out/r4f_dist1.out | cut -c1-400; wc -l gpurun_out/r4f_dist1.out
