#!/bin/bash
# experiments on the matrix-core per-wavelength kernel: pruning levels, one lane
for eps in 0 1e-9 1e-6 1e-3; do
  echo "prune_eps $eps"
  python scripts/variants.py run --cpu-rows 0 --f64-steps 0 --unpruned-steps 0 --streams 1 --steps 100 --prune-eps $eps
done
