#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" python scripts/variants.py run --cpu-rows 24 --f64-steps 0 --unpruned-steps 0 --streams 1 --steps 100 2>&1 | grep default; }
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=6
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=7
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=8
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=4
for p in 6 7; do echo "== clock permax $p"; MPSFR_LIB_PATH=variants/clock.so python scripts/mf2_clock.py $p; done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest3.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r3_pytest3.log
