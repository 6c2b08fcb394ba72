#!/bin/bash
# usage: scripts/r3_trace.sh tag [ENV=val ...] : single-lane kernel trace of bench.py
export TMPDIR=/tmp
tag=$1; shift
OUT=$PWD/gpurun_out/trace_$tag; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --streams 1 > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
f=$(ls $OUT/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    n = r['Name']
    if 'mpsfr' not in n: continue
    short = n.split('::')[-1].split('(')[0][:40]
    if int(r['Calls']) < 20: continue
    print('%-42s calls %4s avg %8.1f us' % (short, r['Calls'], float(r['AverageNs']) / 1e3))
    tot += float(r['AverageNs']) / 1e3
print('sum of per-step kernels: %.1f us' % tot)
PY
