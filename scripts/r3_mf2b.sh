#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" python scripts/variants.py run --cpu-rows 24 --f64-steps 0 --unpruned-steps 0 --streams 1 --steps 100 2>&1 | grep default; }
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=6
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=4
for p in 6 4; do echo "== clock permax $p"; MPSFR_LIB_PATH=variants/clock.so python scripts/mf2_clock.py $p; done
echo "== clock permax 6, mid off"; MPSFR_MF_MID_LOG2=-1000 MPSFR_LIB_PATH=variants/clock.so python scripts/mf2_clock.py 6
