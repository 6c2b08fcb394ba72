#!/bin/bash
# matrix-core per-wavelength path against the FFT path on the other grids
for cfg in "--dim 1280 --rows 100 --nl 35 --steps 20" "--dim 1024 --rows 200 --nl 70 --steps 20" "--dim 256 --rows 100 --nl 35 --npsflin 3 --steps 50" "--dim 128 --rows 100 --nl 35 --steps 100" "--dim 512 --rows 1000 --nl 35 --steps 20"; do
  for m in 1 0; do
    echo "== $cfg otf_mfma=$m"
    MPSFR_OTF_MFMA=$m python scripts/variants.py run --cpu-rows 0 --f64-steps 0 --unpruned-steps 0 $cfg 2>&1 | grep default
  done
done
