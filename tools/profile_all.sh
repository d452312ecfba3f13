#!/bin/bash
# Evidence run for profiles/roundN: rocprofv3 kernel stats + PMC passes of every workload bench.py reports, plus the kernel
# stats of the DEFAULT bench command.  GPU box, repo root:  tools/profile_all.sh [round]  -> gpurun_out/profiles_<round>/
set -u
ROUND=${1:-round4}
R=$PWD
DST=$R/gpurun_out/profiles_$ROUND
mkdir -p $DST
for spec in "direct32x8 1920 1080" "direct32x8 3840 2160" "lut32 1920 1080" "shipped8 1920 1080" "clouds_high 1920 1080" \
            "clouds_high_rm 1920 1080" "clouds_high_rm 3840 2160" "clouds_high@lod0 1920 1080" "clouds_high_rm@lod0 1920 1080" "clouds_high_rm@lod0 3840 2160"; do
  set -- $spec
  tools/profile.sh $1 $2 $3 > /dev/null 2>&1
  python3 tools/summarize_pmc.py gpurun_out/prof_$1_$2x$3 $DST/pmc_$1_$2x$3.json > /dev/null
  cp gpurun_out/prof_$1_$2x$3/stats/s_kernel_stats.csv $DST/kernel_stats_$1_$2x$3.csv 2>/dev/null
done
# the counters just collected are the ones the bench lines below (and every later bench.py of this tree) read: profiles/<round>/pmc_*.json, stamped with this
# build's id -- put them in place BEFORE the default command runs, or its line would say traffic_stale against the previous build's files
mkdir -p $R/profiles/$ROUND && cp $DST/pmc_*.json $DST/kernel_stats_*.csv $R/profiles/$ROUND/ 2>/dev/null
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $DST/default_cmd -o s -- $PY $R/bench.py --also "" > $DST/bench_default_under_rocprof.json 2> $DST/bench_default_under_rocprof.err)
cp $DST/default_cmd/s_kernel_stats.csv $DST/bench_default_command_kernel_stats.csv 2>/dev/null
rm -rf $DST/default_cmd
ATMO_BENCH_DETAIL=$DST/bench_default.json python3 bench.py > $DST/bench_default_line.json 2> $DST/bench_default.err
ls $DST
