for spec in "clouds_high" "clouds_high_rm" "clouds_high P_clouds" "clouds_high_rm P_clouds"; do
  read WL POSE <<< "$spec"; POSE=${POSE:-P_space}
  A=""; B=""
  for r in 1 2 3; do
    for v in 1 2; do
      export ATMO_LANE_SPLIT=$v
      ms=$(python bench.py --workload $WL --pose $POSE --steps 60 --warmup 10 --no-cpu-baseline --also "" 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.readline())['roofline']['kernel_avg_ms'])")
      if [ $v = 1 ]; then A="$A $ms"; else B="$B $ms"; fi
    done
  done
  echo "$WL $POSE  one lane per ray:$A   two lanes per ray:$B"
done
