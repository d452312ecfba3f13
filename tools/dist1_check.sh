export ATMO_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r3h_dist1_k20.out 2> gpurun_out/r3h_dist1_k20.err
python bench.py --steps 200 --warmup 20 > gpurun_out/r3h_dist1_k200.out 2> gpurun_out/r3h_dist1_k200.err
grep -h '^{' gpurun_out/r3h_dist1_k20.out gpurun_out/r3h_dist1_k200.out | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(d['steps'], round(d['value']), c['gather'][:40], round(c['mrays_per_s_no_gather']), round(c['mrays_per_s_final_gather']), round(c['mrays_per_s_gather_every']), list(d.get('extra', {}).get('config4_clouds_high_rm_3840x2160', {}).items())[:8])
"
