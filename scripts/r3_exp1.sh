#!/bin/bash
# experiments: mf2 phase clock; chunk size at the native grid and at 512
mkdir -p gpurun_out
MPSFR_LIB_PATH=variants/clock.so python scripts/mf2_clock.py > gpurun_out/mf2_clock.txt 2>&1; cat gpurun_out/mf2_clock.txt
B="--cpu-rows 0 --f64-steps 0 --profile-steps 10 --unpruned-steps 0 --host-steps 0 --native-steps 0"
one() { python3 bench.py $B "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('   value %.4g PSFs/s  ms_per_step %.4f  kernels %s' % (d['value'], d['ms_per_step'], d['kernel_ms_per_step']))"; }
for ch in 0 8 16 32 64; do echo "== 1280 chunk $ch"; one --dim 1280 --steps 20 --chunk $ch; done
for ch in 0 25 50; do echo "== 512 chunk $ch"; one --steps 200 --chunk $ch; done
for st in 1 2 3; do echo "== 1280 streams $st"; one --dim 1280 --steps 20 --streams $st; done
