#!/bin/bash
# usage: scripts/prof_all.sh <tag>   (run on the GPU box through gpurun)
# 1. rocprofv3 --kernel-trace --stats of the default bench (kernel-time summary)
# 2. separate --pmc passes (no tracing): SQ activity, LDS, and HBM traffic (FETCH_SIZE / WRITE_SIZE)
export TMPDIR=/tmp
R=$PWD
TAG=$1
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 50 --warmup 5 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64 > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
i=0
while read -r P; do
  i=$((i+1))
  D=$OUT/pmc$i
  mkdir -p $D
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 1 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64 > $D/log.txt 2>&1 || { echo "pmc pass $i failed"; tail -5 $D/log.txt; exit 1; }
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE
GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
FETCH_SIZE
WRITE_SIZE
LIST
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
grep '^{' $OUT/trace.log > $OUT/bench_under_rocprof.json
