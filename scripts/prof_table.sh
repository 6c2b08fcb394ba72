#!/bin/bash
# Single-lane profile (no concurrent launches) for the per-kernel resource table of DESIGN.md.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_table
mkdir -p $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64 --streams 1 > $OUT/trace.log 2>&1 || exit 1
i=0
while read -r P; do
  i=$((i+1)); D=$OUT/pmc$i; mkdir -p $D
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 1 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64 --streams 1 > $D/log.txt 2>&1 || exit 1
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES
GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE
LIST
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
