#!/bin/bash
# GPU check of a change: tests, phase clock of the matrix-core kernel (needs variants/clock.so), bench at 1 / 2 / 3 lanes and
# the native grid, single-lane kernel trace.  Run on the GPU box: gpurun -- bash scripts/r3_check.sh
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r3_pytest.log
[ $rc -ne 0 ] && exit 1
MPSFR_LIB_PATH=variants/clock.so python scripts/mf2_clock.py > gpurun_out/mf2_clock.txt 2>&1; cat gpurun_out/mf2_clock.txt | grep -v amdgpu.ids
B="--cpu-rows 48 --f64-steps 0 --profile-steps 10 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64"
one() { python3 bench.py $B "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('   value %.4g  ms_per_step %.4f  parity %s kernels %s sum %.4f it %s' % (d['value'], d['ms_per_step'], d.get('parity',{}).get('max_abs_err_beta'), d['kernel_ms_per_step'], sum(d['kernel_ms_per_step'].values()), d['fit_iterations']))"; }
echo "== default"; one --steps 200
echo "== streams 1"; one --steps 100 --streams 1
echo "== streams 3"; one --steps 200 --streams 3
echo "== native"; one --steps 20 --dim 1280
export TMPDIR=/tmp
OUT=gpurun_out/trace1; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64 --streams 1 > $OUT/trace.log 2>&1
cp $(ls $OUT/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 - <<'PY'
import csv
tot=0
for r in csv.DictReader(open('gpurun_out/trace1/kernel_stats.csv')):
    if int(r['Calls'])<100: continue
    n=r['Name']; import re
    m=re.search(r'k_\w+',n); print('  %-20s calls %5s avg %8.1f us'%(m.group(0) if m else n[:20],r['Calls'],float(r['AverageNs'])/1e3)); tot+=float(r['AverageNs'])/1e3
print('sum %.1f us'%tot)
PY
