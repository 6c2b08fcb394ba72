#!/bin/bash
# usage: scripts/r4_pmc_bench.sh <tag> [bench args]: kernel trace + SQ counter passes of a single-lane bench run
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmcb_$1; shift
rm -rf $OUT; mkdir -p $OUT/trace
B="--cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 64 --streams 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 $B "$@" > $OUT/trace.log 2>&1 || exit 1
i=0
while read -r P; do
  i=$((i+1)); D=$OUT/pmc$i; mkdir -p $D
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 1 $B "$@" > $D/log.txt 2>&1 || exit 1
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES
GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE
LIST
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
grep '^{' $OUT/trace.log > $OUT/bench.json
python3 - $OUT <<'PY'
import csv, sys, re, collections
out = sys.argv[1]
dur = {}
for r in csv.DictReader(open(out + '/kernel_stats.csv')):
    m = re.search(r'(k_\w+(<[^>]*>)?)', r['Name'])
    if m and int(r['Calls']) >= 10: dur[m.group(1)] = float(r['AverageNs']) / 1e3
acc = collections.defaultdict(dict)
cur = None
for ln in open(out + '/pmc_summary.txt'):
    if not ln.startswith(' '): cur = ln.strip()
    else:
        p = ln.split(); acc[cur][p[0]] = float(p[-1].split('=')[1])
print('%-34s %8s %7s %7s %7s %7s %7s %9s %9s' % ('kernel', 'us', 'valu%', 'lds%', 'wait%', 'stall%', 'waves', 'rd MB', 'wr MB'))
for k, us in sorted(dur.items(), key=lambda kv: -kv[1]):
    a = acc.get(k, {})
    cyc = a.get('GRBM_GUI_ACTIVE', 0) / 8 or 1
    wc = a.get('SQ_WAVE_CYCLES', 0) or 1
    print('%-34s %8.1f %7.1f %7.1f %7.1f %7.1f %7.2f %9.1f %9.1f' % (
        k[:34], us, 100 * a.get('SQ_ACTIVE_INST_VALU', 0) * 4 / 1024 / cyc, 100 * a.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / cyc,
        100 * a.get('SQ_WAIT_ANY', 0) / wc, 100 * a.get('SQ_WAIT_INST_ANY', 0) / wc, wc * 4 / 1024 / cyc,
        2 * a.get('FETCH_SIZE', 0) / 1e3, a.get('WRITE_SIZE', 0) / 1e3))
PY
