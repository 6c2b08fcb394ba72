#!/bin/bash
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_colpass
rm -rf $OUT; mkdir -p $OUT
i=0
while read -r P; do
  i=$((i+1)); D=$OUT/p$i; mkdir -p $D
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 1 --cpu-rows 0 > $D/log.txt 2>&1 || { echo fail $i; tail -3 $D/log.txt; }
done <<'LIST'
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAVES
TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
LIST
python3 scripts/pmc_summary.py $OUT 2>/dev/null | grep -A16 "^k_colpass_m" | head -24
