#!/bin/bash
out=gpurun_out/r5_pcopy.txt
: > $out
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0 --steps 300 --warmup 10"
for rep in 1 2; do
for cfg in "1 0 100" "1 1 100" "0 0 100" "0 1 100" "0 0 1000" "0 1 1000" "0 0 25" "0 1 25"; do
  set -- $cfg
  MPSFR_PARAM_COPY=$2 python bench.py $Q --streams $1 --rows $3 > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "FAILED $cfg" >> $out; tail -3 gpurun_out/_l.err >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json'))
print('streams=%s param_copy_kernel=%s rows=%s  %.3f M PSFs/s  ms/step %.4f' % ('$1','$2','$3',b['value']/1e6,b['ms_per_step']))
PY
done; done
cat $out
