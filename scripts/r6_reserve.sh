#!/bin/bash
# VERDICT r5 #1a: the persistent grids of K_OTF_MFMA2 / K_DPHI_SERIES sized to ncu - R workgroups (no CU masks)
# usage: scripts/r6_reserve.sh "<mf a> <mf a> ..." [bench args]
out=gpurun_out/r6_reserve.txt
cfgs=${1:-"0,0 8,8 16,16 32,32 64,64 16,0 32,0 0,16 0,32 0,0"}; shift
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0 --steps 300 --warmup 10"
echo "# bench args: $@" >> $out
for cfg in $cfgs; do
  mf=${cfg%,*}; sa=${cfg#*,}
  MPSFR_PERSIST_RESERVE_MF=$mf MPSFR_PERSIST_RESERVE_A=$sa python bench.py $Q "$@" > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "FAILED $cfg" >> $out; tail -3 gpurun_out/_l.err >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json'))
r=b.get('timed_region_repeats',{})
print('reserve_mf=%s reserve_a=%s  %.3f M PSFs/s  ms/step %.4f  repeats %s' % ('$mf','$sa',b['value']/1e6,b['ms_per_step'],json.dumps(r)[:160]))
PY
done
cat $out
