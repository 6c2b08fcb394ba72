#!/bin/bash
out=gpurun_out/r5_queues2.txt
: > $out
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0 --steps 300 --warmup 10"
for hq in 1 2 3 4; do
for rows in 100 1000; do
  GPU_MAX_HW_QUEUES=$hq python bench.py $Q --rows $rows > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "FAILED hq=$hq rows=$rows" >> $out; tail -3 gpurun_out/_l.err >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json'))
print('GPU_MAX_HW_QUEUES=%d rows=%d  %.3f M PSFs/s  ms/step %.4f' % ($hq,$rows,b['value']/1e6,b['ms_per_step']))
PY
done; done
cat $out
