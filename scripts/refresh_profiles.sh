
bash scripts/prof_table.sh > gpurun_out/prof_table.log 2>&1
python3 scripts/kernel_table.py gpurun_out/prof_table gpurun_out/prof_table/kernel_util.json > gpurun_out/prof_table/kernel_table.md
bash scripts/prof_all.sh r02c > gpurun_out/prof_all.log 2>&1
python3 scripts/traffic_json.py gpurun_out/prof_r02c > gpurun_out/prof_r02c/traffic.json
# native 1280 grid: kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_1280c
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_1280c -- python3 bench.py --dim 1280 --rows 100 --nl 35 --steps 10 --warmup 2 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 > gpurun_out/prof_1280c/trace.log 2>&1
cp $(ls gpurun_out/prof_1280c/*/*kernel_stats.csv | head -1) gpurun_out/prof_1280c/kernel_stats.csv
grep '^{' gpurun_out/prof_1280c/trace.log > gpurun_out/prof_1280c/bench.json
# other configs (builder's run of bench.py, pipelined)
( for cfg in "--rows 1000 --steps 40" "--dim 256 --npsflin 3 --steps 100" "--dim 1024 --rows 200 --nl 70 --steps 20" "--dim 1280 --steps 20" "--dim 128 --steps 200" "--dim 512 --npsflin 3 --rows 50 --steps 40"; do
  echo "== bench.py $cfg"
  python3 bench.py $cfg --cpu-rows 0 --f64-steps 0 --profile-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('   value %.4g PSFs/s  ms_per_step %.4f  unpruned %s  workload: %s' % (d['value'], d['ms_per_step'], d.get('value_unpruned'), d['config']['workload']))"
done ) > gpurun_out/other_configs.txt 2>&1
cat gpurun_out/other_configs.txt
python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
