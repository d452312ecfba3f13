set -x
tools/profile_all.sh round5 > gpurun_out/profile_all_round5.log 2>&1
ATMO_BENCH_DETAIL=gpurun_out/profiles_round5/bench_k20.json python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/profiles_round5/bench_k20_line.json 2> gpurun_out/profiles_round5/bench_k20.err
tail -c 2500 gpurun_out/profiles_round5/bench_k20_line.json
ls gpurun_out/profiles_round5 | head -50
