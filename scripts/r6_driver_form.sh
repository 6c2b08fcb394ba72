#!/bin/bash
# the driver's form (20-step regions from a standing start) several times per setting: first region and the median of the repeats
out=gpurun_out/r6_driver_form.txt
: > $out
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0 --steps 20 --warmup 5"
for rep in 1 2 3; do
for cs in 0 1 2; do
  MPSFR_COLD_STAGGER=$cs python bench.py $Q > gpurun_out/_l.json 2>/dev/null || { echo "FAILED $cs" >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json')); r=b['timed_region_repeats']
print('cold_stagger=$cs  first %.3f  median %.3f  min %.3f  max %.3f' % (b['value']/1e6, r['value_median']/1e6, r['value_min']/1e6, r['value_max']/1e6))
PY
done; done
sort $out
