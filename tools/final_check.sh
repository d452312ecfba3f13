export MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
ATMO_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --shard bands --workload clouds_high_rm > gpurun_out/r3i_bands1.out 2> gpurun_out/r3i_bands1.err; echo "bands rc=$?"
grep -h '^{' gpurun_out/r3i_bands1.out | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print(d['scaling'], round(d['value']), c['shard'][:90], c['gather'][:50], round(c['mrays_per_s_no_gather']), round(c['mrays_per_s_gather_every']))"
unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
T0=$(date +%s); python bench.py > gpurun_out/r3i_default.json 2> gpurun_out/r3i_default.err; echo "default rc=$? wall $(( $(date +%s) - T0 )) s"
python tools/show_bench.py gpurun_out/r3i_default.json 2>&1 | tail -30
