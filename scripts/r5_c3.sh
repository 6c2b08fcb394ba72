#!/bin/bash
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --steps 100 --dim 256 --npsflin 3"
for v in "" variants/oldstart.so; do
for te in 4e-6 inf; do
  MPSFR_LIB_PATH=$v MPSFR_TIER_EPS=$te python bench.py $Q 2>/dev/null | python -c "
import json,sys
b=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('lib=%s tier_eps=%s  %.3f M PSFs/s  ms/step %.4f' % ('$v' or 'default','$te',b['value']/1e6,b['ms_per_step']), {k:round(v,4) for k,v in b['kernel_ms_one_call_in_flight'].items()}, b['fit_iterations'])"
done; done
