#!/bin/bash
# Round-6 profile refresh (run on the GPU box through gpurun): single-lane table + PMC (configs[1]), two-lane
# trace + PMC + traffic, native 1280 and f64 single-lane tables with PMC, the fp64 issue-rate micro-benchmark.
# Outputs under gpurun_out/; scripts/r6_collect.sh copies the summaries into profiles/r06_*.
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/prof_table.sh > gpurun_out/prof_table.log 2>&1 || { echo prof_table failed; tail -5 gpurun_out/prof_table.log; exit 1; }
python3 scripts/kernel_table.py gpurun_out/prof_table gpurun_out/prof_table/kernel_util.json > gpurun_out/prof_table/kernel_table.md
echo "table done"
bash scripts/prof_all.sh r06 > gpurun_out/prof_all.log 2>&1 || { echo prof_all failed; tail -5 gpurun_out/prof_all.log; exit 1; }
python3 scripts/traffic_json.py gpurun_out/prof_r06 > gpurun_out/prof_r06/traffic.json
echo "two-lane done"
scripts/r4_pmc_bench.sh native1280 --dim 1280 --rows 100 --nl 35 > gpurun_out/pmcb_native1280.txt 2>&1 || { echo native failed; tail -5 gpurun_out/pmcb_native1280.txt; exit 1; }
echo "native done"
scripts/r4_pmc_bench.sh f64 --precision f64 > gpurun_out/pmcb_f64.txt 2>&1 || { echo f64 failed; tail -5 gpurun_out/pmcb_f64.txt; exit 1; }
echo "f64 done"
scripts/r4_pmc_bench.sh c1 > gpurun_out/pmcb_c1.txt 2>&1 || { echo c1 failed; exit 1; }
scripts/ubench/dpp64 > gpurun_out/ubench_dpp64.txt 2>&1
echo "all done"
