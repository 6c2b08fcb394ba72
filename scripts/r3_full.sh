#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest5.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r3_pytest5.log
timeout -k 10 600 python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?"; tail -3 gpurun_out/r3_bench1.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r3_bench1.json') if l.startswith('{')][0])
for k in ('value', 'ms_per_step', 'value_host_outputs', 'value_native1280', 'value_f64', 'value_unpruned'):
    print(k, d.get(k))
print('roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'avg_launch_ms')}, d['roofline']['one_call_in_flight'])
print('parity', d.get('parity')); print('native', d.get('native1280')); print('host', d.get('host_outputs'))
print('repeats', d.get('timed_region_repeats')); print('unpruned', d.get('unpruned'))
print('kernels', d['kernel_ms_per_step'])
PY
