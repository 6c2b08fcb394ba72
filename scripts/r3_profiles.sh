#!/bin/bash
# Round-3 profile refresh (run on the GPU box through gpurun): single-lane table + PMC, two-lane trace + PMC
# + traffic, native 1280 trace + PMC.  Outputs under gpurun_out/; copy the summaries into profiles/r03_*.
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/prof_table.sh > gpurun_out/prof_table.log 2>&1 || { echo prof_table failed; tail -5 gpurun_out/prof_table.log; exit 1; }
python3 scripts/kernel_table.py gpurun_out/prof_table gpurun_out/prof_table/kernel_util.json > gpurun_out/prof_table/kernel_table.md
echo "table done"
bash scripts/prof_all.sh r03 > gpurun_out/prof_all.log 2>&1 || { echo prof_all failed; tail -5 gpurun_out/prof_all.log; exit 1; }
python3 scripts/traffic_json.py gpurun_out/prof_r03 > gpurun_out/prof_r03/traffic.json
echo "two-lane done"
# native 1280 grid: kernel trace (single lane) and a PMC pass
OUT=gpurun_out/prof_1280; mkdir -p $OUT/trace
NAT="--dim 1280 --rows 100 --nl 35 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64 --streams 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $NAT --steps 10 --warmup 2 > $OUT/trace.log 2>&1 || { echo native trace failed; tail -5 $OUT/trace.log; exit 1; }
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
grep '^{' $OUT/trace.log > $OUT/bench.json
i=0
while read -r P; do
  i=$((i+1)); D=$OUT/pmc$i; mkdir -p $D
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $D -- python3 bench.py $NAT --steps 2 --warmup 1 > $D/log.txt 2>&1 || { echo "native pmc $i failed"; tail -5 $D/log.txt; exit 1; }
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES
FETCH_SIZE
WRITE_SIZE
LIST
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
echo "native done"
