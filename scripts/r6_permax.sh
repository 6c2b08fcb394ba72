#!/bin/bash
# wavelengths per workgroup of K_OTF_MFMA2 (fewer: 2 waves per SIMD and 112 KB of LDS instead of 3 and 136 KB, i.e. room for
# another kernel's waves on the same CU) x the reserve of the persistent grids
out=gpurun_out/r6_permax.txt
: > $out
Q="--f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --cpu-rows 0 --steps 300 --warmup 10"
for pm in 6 5 4 3 7; do
for r in -1 0; do
  MPSFR_MF_PERMAX=$pm MPSFR_PERSIST_RESERVE=$r python bench.py $Q "$@" > gpurun_out/_l.json 2>/dev/null || { echo "FAILED $pm $r" >> $out; continue; }
  python - >> $out <<PY
import json
b=json.load(open('gpurun_out/_l.json')); k=b['kernel_ms_one_call_in_flight']
print('mf_permax=$pm reserve=%2s  %.3f M PSFs/s  ms/step %.4f | alone: otf_mfma %.4f ms' % ('$r', b['value']/1e6, b['ms_per_step'], k.get('otf_mfma', 0)))
PY
done; done
cat $out
