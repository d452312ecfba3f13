#!/bin/bash
# Calibrates the SQ instruction-class counters of rocprofv3 on the single-opcode kernels of tools/valu_issue: every
# kernel there issues a known number of one VALU opcode, so the CSVs say which counter each opcode increments and how
# many "active" quad-cycles the SQ books for it.  PMC passes only (never combined with tracing).
# Usage (GPU box, repo root):  tools/calibrate_counters.sh  ->  gpurun_out/calib/pass_*/..._counter_collection.csv
set -u
R=$PWD
OUT=$R/gpurun_out/calib
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for PMC in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for SET in set1 set2; do
    rocprofv3 --pmc $PMC --output-format csv -d $OUT/pass_${i}_$SET -o c -- $R/tools/valu_issue $SET > $OUT/pass_${i}_$SET.log 2>&1
  done
done
ls $OUT
