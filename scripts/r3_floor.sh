#!/bin/bash
# round 3, first GPU call: the test suite, then the fp16 representation floor on the old kernel
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest0.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_pytest0.log
for f in 1 0; do
  echo "== mf_floor=$f, one lane"
  MPSFR_MF_FLOOR=$f python scripts/variants.py run --cpu-rows 0 --f64-steps 0 --streams 1 --steps 100 2>&1 | grep default
done
echo "== default bench (two lanes), floor on"
python bench.py --cpu-rows 24 --steps 200 > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err; tail -c 3000 gpurun_out/r3_bench0.json
