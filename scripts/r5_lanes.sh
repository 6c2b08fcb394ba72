#!/bin/bash
# CU-partitioned lanes: lanes x cu_partition at configs[1] (100 rows) and configs[2] on one GPU (1000 rows)
out=gpurun_out/r5_lanes.txt
: > $out
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0 --steps 300 --warmup 10"
for rows in 100 1000; do
for st in 2 3 4; do
for cp in 0 1; do
  MPSFR_CU_PARTITION=$cp python bench.py $Q --rows $rows --streams $st > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "FAILED rows=$rows streams=$st cu_partition=$cp" >> $out; tail -3 gpurun_out/_l.err >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json'))
print('rows=%d streams=%d cu_partition=%d  %.3f M PSFs/s  ms/step %.4f  (min %.3f max %.3f)' % ($rows,$st,$cp,b['value']/1e6,b['ms_per_step'],b.get('value_min',0)/1e6,b.get('value_max',0)/1e6))
PY
done; done; done
cat $out
