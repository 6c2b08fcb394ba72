#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest4.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 gpurun_out/r3_pytest4.log
bash scripts/r3_trace.sh p6b MPSFR_MF_PERMAX=6
run() { echo "== $*"; env "$@" python scripts/variants.py run --cpu-rows 24 --f64-steps 0 --unpruned-steps 0 --steps 200 2>&1 | grep default; }
run MPSFR_MF_PERMAX=6
run MPSFR_MF_PERMAX=8
run MPSFR_MF_KERNEL=1
