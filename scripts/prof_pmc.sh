#!/bin/bash
# usage: scripts/prof_pmc.sh <outdir-under-gpurun_out> <bench args...>
# Runs rocprofv3 --pmc passes (counters only, no tracing) of bench.py and summarises per kernel.
# --cpu-rows 0 is forced: the CPU-baseline leg forks a pool, and rocprofv3 has initialised the GPU
# in the parent before bench.py starts.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
i=0
while read -r P; do
  i=$((i+1))
  D=$OUT/pass$i
  mkdir -p $D
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $D -- python3 bench.py "$@" --cpu-rows 0 --prime 64 > $D/log.txt 2>&1 || { echo "pass $i failed"; tail -5 $D/log.txt; exit 1; }
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE
GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
LIST
python3 scripts/pmc_summary.py $OUT > $OUT/summary.txt
