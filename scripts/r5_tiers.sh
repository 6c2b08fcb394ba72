#!/bin/bash
# potential of higher tier thresholds (no budget): floor / mid sweeps at configs[1]
out=gpurun_out/r5_tiers.txt
: > $out
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 12 --steps 300 --warmup 10"
for cfg in "-29.01 -18.01" "-26 -18.01" "-23 -18.01" "-20 -18.01" "-29.01 -14" "-29.01 -10" "-23 -12" "-20 -10" "-18 -8"; do
  set -- $cfg
  MPSFR_TIER_EPS=inf MPSFR_MF_FLOOR_LOG2=$1 MPSFR_MF_MID_LOG2=$2 python bench.py $Q > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "FAILED $cfg" >> $out; tail -3 gpurun_out/_l.err >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json'))
k=b.get('kernel_ms_one_call_in_flight',{})
p=b.get('parity') or {'max_abs_err_fwhm_arcsec': float('nan'), 'max_abs_err_beta': float('nan')}
print('floor %s mid %s  %.3f M PSFs/s  otf_mfma alone %.1f us  prep %.1f  steps frac %.4f | parity fwhm %.2e beta %.2e' % ('$1','$2',b['value']/1e6,1e3*k.get('otf_mfma',0),1e3*k.get('mf_prep',0),b['roofline'].get('tile_steps_executed_fraction',0),p['max_abs_err_fwhm_arcsec'],p['max_abs_err_beta']))
PY
done
cat $out
