#!/bin/bash
# f64 mode: single-lane kernel table
export TMPDIR=/tmp
OUT=gpurun_out/trace_f64; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --precision f64 --steps 10 --warmup 2 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64 --streams 1 > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
cp $(ls $OUT/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 - <<'PY'
import csv,re
tot=0
for r in csv.DictReader(open('gpurun_out/trace_f64/kernel_stats.csv')):
    if int(r['Calls'])<50: continue
    m=re.search(r'k_\w+(<[^>]*>)?',r['Name']); print('  %-40s calls %5s avg %8.1f us'%(m.group(0) if m else r['Name'][:30],r['Calls'],float(r['AverageNs'])/1e3)); tot+=float(r['AverageNs'])/1e3
print('sum %.1f us'%tot)
PY
grep '^{' $OUT/trace.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms', d['ms_per_step'], d['fit_iterations'], d['roofline']['lines_transformed_fraction'])"
