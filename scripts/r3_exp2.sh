#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest6.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r3_pytest6.log
[ $rc -ne 0 ] && exit 1
python scripts/variants.py run --cpu-rows 48 --f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --steps 200 2>&1 | grep -v amdgpu.ids
MPSFR_LIB_PATH=variants/clock.so python scripts/mf2_clock.py > gpurun_out/mf2_clock2.txt 2>&1; cat gpurun_out/mf2_clock2.txt | grep -v amdgpu.ids
B="--cpu-rows 0 --f64-steps 0 --profile-steps 10 --unpruned-steps 0 --host-steps 0 --native-steps 0"
python3 bench.py $B --steps 100 --streams 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('single lane: value %.4g  ms_per_step %.4f  kernels %s sum %.4f' % (d['value'], d['ms_per_step'], d['kernel_ms_per_step'], sum(d['kernel_ms_per_step'].values())))"
python3 bench.py $B --steps 20 --dim 1280 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('native: value %.4g  ms_per_step %.4f  kernels %s' % (d['value'], d['ms_per_step'], d['kernel_ms_per_step']))"
