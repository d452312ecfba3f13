#!/bin/bash
# HBM read traffic (FETCH_SIZE, its own PMC pass) of one workload with a switch on / off:  tools/fetch_check.sh VAR workload [W H]
set -u
VAR=$1; WL=$2; W=${3:-1920}; H=${4:-1080}
R=$PWD
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
export TMPDIR=/tmp
cd /tmp
for v in on off; do
  if [ $v = off ]; then export $VAR=${OFF:-0}; else unset $VAR; fi
  OUT=$R/gpurun_out/fetch_${WL}_${W}x${H}_$v
  rm -rf $OUT; mkdir -p $OUT
  for PMC in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TCP_TCC_READ_REQ_sum"; do
    d=$OUT/pmc_$(echo $PMC | tr ' ' '_')
    rocprofv3 --pmc $PMC --output-format csv -d $d -o p -- $PY $R/bench.py --workload $WL --width $W --height $H --no-cpu-baseline --also , --steps 6 --warmup 2 > $d.log 2>&1
  done
  python3 $R/tools/summarize_pmc.py $OUT $OUT/summary.json > /dev/null 2>&1
  python3 - <<PYEOF
import json
d = json.load(open("$OUT/summary.json"))
p = d.get("pmc_per_launch", {})
print("$WL ${W}x$H $VAR", "$v", {k: round(v["mean_per_launch"]) for k, v in p.items()}, "HBM MB per launch", round(d.get("derived", {}).get("hbm_bytes_per_launch", 0) / 1e6, 2), "L2 hit", round(d.get("derived", {}).get("l2_hit_rate", 0), 4))
PYEOF
done
