#!/bin/bash
out=gpurun_out/r6_driver_form2.txt
: > $out
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0 --steps 20 --warmup 5"
for rep in 1 2 3 4; do
for r in -1 32; do
  MPSFR_PERSIST_RESERVE=$r python bench.py $Q > gpurun_out/_l.json 2>/dev/null || { echo "FAILED $r" >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json')); r=b['timed_region_repeats']
print('persist_reserve=%3s  first %.3f  median %.3f  min %.3f  max %.3f' % ('$r', b['value']/1e6, r['value_median']/1e6, r['value_min']/1e6, r['value_max']/1e6))
PY
done; done
sort $out
