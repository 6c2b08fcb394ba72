#!/bin/bash
# usage: scripts/r4_trace.sh <tag> [bench args]: kernel trace of a single-lane bench run, per-kernel averages
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace_$1; shift
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 3 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64 --streams 1 "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
cp $(ls $OUT/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    m = re.search(r'(k_\w+(<[^>]*>)?)', r['Name'])
    name = m.group(1) if m else r['Name'][:40]
    calls = int(r['Calls'])
    if calls < 10: continue
    avg = float(r['AverageNs']) / 1e3
    tot += avg
    print('%-44s calls %4d  avg %8.1f us' % (name, calls, avg))
print('sum of per-call kernels %.1f us' % tot)
PY
