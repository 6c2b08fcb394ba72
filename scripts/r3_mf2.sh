#!/bin/bash
# round 3: the thin-wave matrix-core kernel against the old one (one lane, kernel times from the event pass)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest1.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3_pytest1.log
run() { echo "== $*"; env "$@" python scripts/variants.py run --cpu-rows 24 --f64-steps 0 --unpruned-steps 0 --streams 1 --steps 100 2>&1 | grep default; }
run MPSFR_MF_KERNEL=1
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=6
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=6 MPSFR_MF_MID_LOG2=-1000
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=5
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=4
run MPSFR_MF_KERNEL=2 MPSFR_MF_PERMAX=7
