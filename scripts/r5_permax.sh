#!/bin/bash
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --profile-steps 0 --steps 300 --warmup 10"
for rep in 1 2; do
for pm in 6 5 4 7; do
  MPSFR_MF_PERMAX=$pm python bench.py $Q 2>/dev/null | python -c "
import json,sys
b=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('mf_permax=$pm  %.3f M PSFs/s  ms/step %.4f' % (b['value']/1e6,b['ms_per_step']))"
done; done
