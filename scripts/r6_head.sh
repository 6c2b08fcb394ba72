#!/bin/bash
# round 6: head fusion (blob copy in K_PATCH_GEN, kernel spectra in K_PATCH_ROWS) x persist_reserve
out=gpurun_out/r6_head.txt
: > $out
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0 --steps 300 --warmup 10"
for args in "" "--streams 1" "--rows 1000 --steps 60" "--rows 25" "--dim 1280 --steps 100"; do
echo "# bench args: $args" >> $out
CFGS=${CFGS:-0,0 1,0 0,32 1,32 1,32}
for cfg in $CFGS; do cfg=${cfg//,/ }
  set -- $cfg
  MPSFR_HEAD_FUSION=$1 MPSFR_PERSIST_RESERVE=$2 MPSFR_COPY_FUSION=${3:-1} timeout -k 10 120 python bench.py $Q $args > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "FAILED $cfg" >> $out; tail -3 gpurun_out/_l.err >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json'))
r=b.get('timed_region_repeats',{})
print('head_fusion=%s reserve=%s copy_fusion=${3:-1}  %.3f M PSFs/s  ms/step %.4f  median %.3f' % ('$1','$2',b['value']/1e6,b['ms_per_step'],r.get('value_median',0)/1e6))
PY
done; done
cat $out
