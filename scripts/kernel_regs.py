"""Register / LDS / scratch use of every kernel of a .hip source (compiles it to gfx950 assembly).
usage: python scripts/kernel_regs.py muse_psfr_amd/csrc/per_lambda.hip [filter]"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
with tempfile.NamedTemporaryFile(suffix='.s') as f:
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17',
                    '-fno-slp-vectorize', '-x', 'hip', '--cuda-device-only', '-S', '-o', f.name, src] +
                   sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
    txt = open(f.name).read()
for blk in re.split(r'\n  - \.agpr_count', txt)[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = subprocess.run(['c++filt', g('name')], capture_output=True,
                          text=True).stdout.strip()
    name = name.replace('mpsfr::(anonymous namespace)::', '').split('(')[0]
    if flt in name:
        print('%-60s vgpr %4s sgpr %4s scratch %5s lds %6s' % (name[:60], g('vgpr_count'), g('sgpr_count'),
              g('private_segment_fixed_size'), g('group_segment_fixed_size')))
