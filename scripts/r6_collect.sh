#!/bin/bash
# copy the summaries of scripts/r5_profiles.sh (gpurun_out/) into profiles/r06_*
set -e
P=profiles
cp gpurun_out/prof_table/kernel_stats.csv $P/r06_kernel_stats_single_lane.csv
cp gpurun_out/prof_table/pmc_summary.txt $P/r06_pmc_summary_single_lane.txt
cp gpurun_out/prof_table/kernel_table.md $P/r06_kernel_table.md
cp gpurun_out/prof_table/kernel_util.json $P/r06_kernel_util.json
cp gpurun_out/prof_r06/kernel_stats.csv $P/r06_kernel_stats.csv
cp gpurun_out/prof_r06/pmc_summary.txt $P/r06_pmc_summary.txt
cp gpurun_out/prof_r06/traffic.json $P/r06_traffic.json
cp gpurun_out/prof_r06/bench_under_rocprof.json $P/r06_bench_under_rocprof.json
cp gpurun_out/pmcb_native1280/kernel_stats.csv $P/r06_kernel_stats_native1280.csv
cp gpurun_out/pmcb_native1280/pmc_summary.txt $P/r06_pmc_native1280.txt
cp gpurun_out/pmcb_native1280.txt $P/r06_kernel_table_native1280.txt
cp gpurun_out/pmcb_native1280/bench.json $P/r06_bench_under_rocprof_native1280.json
cp gpurun_out/pmcb_f64/kernel_stats.csv $P/r06_kernel_stats_f64.csv
cp gpurun_out/pmcb_f64/pmc_summary.txt $P/r06_pmc_f64.txt
cp gpurun_out/pmcb_f64.txt $P/r06_kernel_table_f64.txt
cp gpurun_out/pmcb_c1.txt $P/r06_kernel_table_c1.txt
cp gpurun_out/ubench_dpp64.txt $P/r06_ubench_dpp64.txt
[ -f gpurun_out/parity_margins.json ] && cp gpurun_out/parity_margins.json $P/r06_parity_margins.json
[ -f gpurun_out/other_configs.txt ] && cp gpurun_out/other_configs.txt $P/r06_other_configs.txt
[ -f gpurun_out/bench_final.json ] && cp gpurun_out/bench_final.json $P/r06_bench_builder_run.json
ls -la $P/r06_*
