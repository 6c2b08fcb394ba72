#!/bin/bash
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0"
for rep in 1 2; do
for cs in 0 1 2; do
  MPSFR_COLD_STAGGER=$cs python bench.py $Q 2>/dev/null | python -c "
import json,sys
b=json.loads([l for l in sys.stdin if l.startswith('{')][0])
r=b.get('timed_region_repeats',{})
print('cold_stagger=$cs  %.3f M PSFs/s  ms/step %.4f  repeats median %.3f M' % (b['value']/1e6,b['ms_per_step'],r.get('value_median',0)/1e6))"
  MPSFR_COLD_STAGGER=$cs python bench.py $Q --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
b=json.loads([l for l in sys.stdin if l.startswith('{')][0])
r=b.get('timed_region_repeats',{})
print('   20 steps: cold_stagger=$cs  %.3f M PSFs/s  repeats median %.3f min %.3f max %.3f' % (b['value']/1e6,r.get('value_median',0)/1e6,r.get('value_min',0)/1e6,r.get('value_max',0)/1e6))"
done; done
