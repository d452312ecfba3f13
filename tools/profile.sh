#!/bin/bash
# rocprofv3 evidence for one bench workload: kernel-trace stats + separate PMC passes (never combined with
# tracing).  Usage (on the GPU box, from the repo root):  tools/profile.sh <workload> [width height]
# Output: gpurun_out/prof_<workload>/{stats,pmc_*}/...csv ; summarise with tools/summarize_pmc.py
set -u
WL=${1:-direct32x8}; W=${2:-1920}; H=${3:-1080}; POSE=${4:-P_space}
SAMPLER=declared; case "$WL" in *@lod0) SAMPLER=lod0; WLN=${WL%@lod0};; *) WLN=$WL;; esac   # clouds_high@lod0 = --workload clouds_high --sampler lod0
R=$PWD
# the real interpreter binary, resolved BEFORE profiling: a pyenv/conda shim or wrapper script after `--` would be an
# exec hop behind the profiler's preloaded (GPU-initialising) library, which this pool forbids
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
OUT=$R/gpurun_out/prof_${WL}_${W}x${H}$( [ "$POSE" = P_space ] || echo _$POSE )
mkdir -p $OUT
# which kernels these counters belong to (bench.py compares it with the library it is timing: traffic_stale)
$PY -c "import sys; sys.path.insert(0, '$R'); from godot_atmosphere_shader_amd import _native as N; print(N.load().atmo_build_id().decode())" > $OUT/build_id.txt 2>/dev/null
export TMPDIR=/tmp
cd /tmp
ARGS="--workload $WLN --sampler $SAMPLER --width $W --height $H --pose $POSE --no-cpu-baseline --also ,"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $PY $R/bench.py $ARGS --steps 50 --warmup 5 > $OUT/stats.log 2>&1
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F16 SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "VALUBusy" "VALUUtilization" "OccupancyPercent" "MemUnitStalled"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc_$i -o p -- $PY $R/bench.py $ARGS --steps 6 --warmup 2 > $OUT/pmc_$i.log 2>&1
done
ls $OUT
