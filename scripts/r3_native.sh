#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace_native; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --dim 1280 --steps 10 --warmup 2 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --streams 1 > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
f=$(ls $OUT/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    n = r['Name']
    if 'mpsfr' not in n or int(r['Calls']) < 20: continue
    short = n.split('::')[-1].split('(')[0][:40]
    print('%-42s calls %4s avg %8.1f us' % (short, r['Calls'], float(r['AverageNs']) / 1e3))
    tot += float(r['AverageNs']) / 1e3
print('sum of per-step kernels: %.1f us' % tot)
PY
grep '^{' $OUT/trace.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms', d['ms_per_step'], d['roofline']['tile_steps_executed_fraction'])"
