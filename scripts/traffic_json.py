"""HBM traffic per kernel from the FETCH_SIZE / WRITE_SIZE passes of scripts/prof_all.sh.
usage: python scripts/traffic_json.py gpurun_out/prof_<tag> > profiles/rNN_traffic.json"""
import json
import re
import sys

txt = open(sys.argv[1] + '/pmc_summary.txt').read()
kern = {}
for blk in re.split(r'\n(?=\S)', txt):
    name = blk.split('\n')[0].strip()
    m = re.match(r'(k_\w+)', name)
    if not m:
        continue
    f = re.search(r'FETCH_SIZE\s+n=\s*(\d+)\s+mean=([0-9.e+]+)', blk)
    w = re.search(r'WRITE_SIZE\s+n=\s*(\d+)\s+mean=([0-9.e+]+)', blk)
    if f and w and int(f.group(1)) > 2:
        key = 'k_otf_rowfft' if m.group(1) == 'k_otf_r16' else m.group(1)
        kern[key] = {'fetch_kib': float(f.group(2)), 'write_kib': float(w.group(2)), 'symbol': name}
print(json.dumps({
    'round': 6,
    'command': 'scripts/prof_all.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, no tracing) '
               '-- python3 bench.py --steps 3 --warmup 1 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0',
    'workload': '100 rows x 35 lambda x 512^2, mixed precision, one context with two lanes, one launch of '
                '100 rows per step',
    'units': 'FETCH_SIZE/WRITE_SIZE in KiB per launch (rocprofv3), mean over launches; FETCH_SIZE is not '
             'corrected (the 2x gfx950 correction of the guide is calibrated for 16 B/lane streams only)',
    'units_per_launch': 3500,
    'kernels': kern}, indent=1))
