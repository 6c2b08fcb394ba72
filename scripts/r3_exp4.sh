#!/bin/bash
# native-grid stage A: lines per workgroup (variants built by scripts/variants.py)
export TMPDIR=/tmp
mkdir -p gpurun_out
NAT="--dim 1280 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0"
for lib in default $(ls variants/*.so); do
  if [ $lib != default ]; then export MPSFR_LIB_PATH=$lib; fi
  OUT=gpurun_out/nat_$(basename $lib .so); mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py $NAT --steps 10 --warmup 2 --streams 1 > $OUT/trace.log 2>&1 || { echo failed $lib; tail -3 $OUT/trace.log; continue; }
  f=$(ls $OUT/*/*kernel_stats.csv | head -1)
  v2=$(python3 bench.py $NAT --steps 20 2>/dev/null | python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][0])['value'])")
  python3 - $f $lib $v2 <<'PY'
import csv,sys,re
d={}
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r'k_\w+',r['Name'])
    if m and int(r['Calls'])>50: d[m.group(0)]=float(r['AverageNs'])/1e3
print('%-18s psd_rowfft %7.1f colfft %7.1f mfma2 %6.1f dmin %5.1f sum %7.1f | two lanes: %s PSFs/s' % (sys.argv[2][-14:], d.get('k_psd_rowfft',0), d.get('k_colfft_dphi',0), d.get('k_otf_mfma2',0), d.get('k_dmin',0), sum(d.values()), sys.argv[3]))
PY
done
