R=$PWD; DST=$R/gpurun_out/profiles_round3; mkdir -p $DST
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $DST/default_cmd -o s -- $PY $R/bench.py > $DST/bench_default_under_rocprof.json 2> $DST/bench_default_under_rocprof.err)
cp $DST/default_cmd/s_kernel_stats.csv $DST/bench_default_command_kernel_stats.csv
rm -rf $DST/default_cmd
head -12 $DST/bench_default_command_kernel_stats.csv | cut -c1-150
