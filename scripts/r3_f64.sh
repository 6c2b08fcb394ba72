#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest6.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r3_pytest6.log
echo "== f64 two lanes"; python scripts/variants.py run --precision f64 --cpu-rows 24 --f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --steps 40 2>&1 | grep default
echo "== f64 one lane"; python scripts/variants.py run --precision f64 --cpu-rows 0 --f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --steps 40 --streams 1 2>&1 | grep default
