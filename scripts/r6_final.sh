#!/bin/bash
# Round-6 final evidence run (on the GPU box through gpurun): the whole GPU test suite (parity margins), the default
# bench (builder's run), the driver's form, the other BASELINE configurations, the host paths, the robustness sweep.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gputest_final.txt 2>&1; tail -3 gpurun_out/r6_gputest_final.txt
cp gpurun_out/parity_margins.json gpurun_out/parity_margins_final.json
python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { echo bench failed; tail -5 gpurun_out/bench_final.err; exit 1; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_form.json 2> gpurun_out/bench_driver_form.err || { echo driver form failed; exit 1; }
bash scripts/r3_other_configs.sh > /dev/null 2>&1
timeout -k 10 600 python3 scripts/robust_sweep.py > gpurun_out/robust_sweep.txt 2>&1; tail -2 gpurun_out/robust_sweep.txt
echo done
