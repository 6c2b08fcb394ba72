#!/bin/bash
# hardware queues: HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues
out=gpurun_out/r5_queues.txt
: > $out
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0 --steps 300 --warmup 10"
for hq in 4 8 16; do
for cfg in "0 0" "0 1" "0 3" "3 0" "4 0"; do
  set -- $cfg
  st=$1; ax=$2
  GPU_MAX_HW_QUEUES=$hq MPSFR_AUX=$ax python bench.py $Q --streams $st > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "FAILED hq=$hq streams=$st aux=$ax" >> $out; tail -3 gpurun_out/_l.err >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json'))
print('GPU_MAX_HW_QUEUES=%d streams=%d aux=%d  %.3f M PSFs/s  ms/step %.4f' % ($hq,$st,$ax,b['value']/1e6,b['ms_per_step']))
PY
done; done
cat $out
