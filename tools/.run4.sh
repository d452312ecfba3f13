mkdir -p gpurun_out/r5d
(time python -m pytest tests -m gpu -q -s 2>&1) > gpurun_out/r5d/pytest.log 2>&1
tail -5 gpurun_out/r5d/pytest.log
for K in 20 200; do for rep in 1 2; do for ev in 0 1; do
  echo -n "K=$K ATMO_DRAW_EVENTS=$ev: "
  ATMO_DRAW_EVENTS=$ev ATMO_BENCH_DETAIL= python bench.py --steps $K --warmup 10 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done; done; done > gpurun_out/r5d/ab_draw_events_home.txt 2>&1
cat gpurun_out/r5d/ab_draw_events_home.txt
