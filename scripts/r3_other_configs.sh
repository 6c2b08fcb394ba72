#!/bin/bash
# builder's run of bench.py on the other BASELINE configurations (pipelined, two lanes)
mkdir -p gpurun_out
( for cfg in "--rows 1000 --steps 40" "--dim 256 --npsflin 3 --steps 100" "--dim 1024 --rows 200 --nl 70 --steps 20" "--dim 1280 --steps 20" "--dim 128 --steps 200" "--dim 512 --npsflin 3 --rows 50 --steps 40" "--precision f64 --steps 50"; do
  echo "== bench.py $cfg"
  python3 bench.py $cfg --cpu-rows 0 --f64-steps 0 --profile-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('   value %.4g PSFs/s  ms_per_step %.4f  unpruned %s  workload: %s' % (d['value'], d['ms_per_step'], d.get('value_unpruned'), d['config']['workload']))"
done ) > gpurun_out/other_configs.txt 2>&1
cat gpurun_out/other_configs.txt
python3 scripts/host_path_rate.py > gpurun_out/host_paths.txt 2>&1; cat gpurun_out/host_paths.txt | grep -v amdgpu.ids
