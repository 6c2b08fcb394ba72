#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest2.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 gpurun_out/r3_pytest2.log
[ $rc -ne 0 ] && exit 1
run() { echo "== $*"; env "$@" python scripts/variants.py run --cpu-rows 24 --f64-steps 0 --unpruned-steps 0 --streams 1 --steps 100 2>&1 | grep default; }
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=6
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=4
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=3
for p in 6 4; do echo "== clock permax $p"; MPSFR_LIB_PATH=variants/clock.so python scripts/mf2_clock.py $p; done
