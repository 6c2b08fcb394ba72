#!/bin/bash
# copy the summaries of scripts/r4_profiles.sh (gpurun_out/) into profiles/r04_*
set -e
P=profiles
cp gpurun_out/prof_table/kernel_stats.csv $P/r04_kernel_stats_single_lane.csv
cp gpurun_out/prof_table/pmc_summary.txt $P/r04_pmc_summary_single_lane.txt
cp gpurun_out/prof_table/kernel_table.md $P/r04_kernel_table.md
cp gpurun_out/prof_table/kernel_util.json $P/r04_kernel_util.json
cp gpurun_out/prof_r04/kernel_stats.csv $P/r04_kernel_stats.csv
cp gpurun_out/prof_r04/pmc_summary.txt $P/r04_pmc_summary.txt
cp gpurun_out/prof_r04/traffic.json $P/r04_traffic.json
cp gpurun_out/prof_r04/bench_under_rocprof.json $P/r04_bench_under_rocprof.json
cp gpurun_out/pmcb_native1280/kernel_stats.csv $P/r04_kernel_stats_native1280.csv
cp gpurun_out/pmcb_native1280/pmc_summary.txt $P/r04_pmc_native1280.txt
cp gpurun_out/pmcb_native1280.txt $P/r04_kernel_table_native1280.txt
cp gpurun_out/pmcb_native1280/bench.json $P/r04_bench_under_rocprof_native1280.json
cp gpurun_out/pmcb_f64/kernel_stats.csv $P/r04_kernel_stats_f64.csv
cp gpurun_out/pmcb_f64/pmc_summary.txt $P/r04_pmc_f64.txt
cp gpurun_out/pmcb_f64.txt $P/r04_kernel_table_f64.txt
cp gpurun_out/pmcb_c1.txt $P/r04_kernel_table_c1.txt
cp gpurun_out/ubench_dpp64.txt $P/r04_ubench_dpp64.txt
[ -f gpurun_out/parity_margins.json ] && cp gpurun_out/parity_margins.json $P/r04_parity_margins.json
[ -f gpurun_out/other_configs.txt ] && cp gpurun_out/other_configs.txt $P/r04_other_configs.txt
[ -f gpurun_out/bench_final.json ] && cp gpurun_out/bench_final.json $P/r04_bench_builder_run.json
ls -la $P/r04_*
