#!/bin/bash
# usage: scripts/r4_pmc.sh <tag> <python script and args...>: kernel trace + three SQ counter passes of one script
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_$1; shift
rm -rf $OUT; mkdir -p $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2>&1 || exit 1
i=0
while read -r P; do
  i=$((i+1)); D=$OUT/pmc$i; mkdir -p $D
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $D -- python3 "$@" > $D/log.txt 2>&1 || exit 1
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
GRBM_GUI_ACTIVE
LIST
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
cut -c1-160 $OUT/kernel_stats.csv | head -12
