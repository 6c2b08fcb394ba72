"""Per-kernel resource table from scripts/prof_table.sh output (single lane, kernels do not overlap).
usage: python scripts/kernel_table.py gpurun_out/prof_table > profiles/rNN_kernel_table.md"""
import csv
import json
import re
import sys

d = sys.argv[1]
txt = open(d + '/pmc_summary.txt').read()
blocks = {b.split('\n')[0]: b for b in re.split(r'\n(?=\S)', txt)}


def get(k, c):
    for name, b in blocks.items():
        if name.startswith(k):
            m = re.search(c + r'\s+n=\s*\d+\s+mean=([0-9.e+]+)', b)
            if m:
                return float(m.group(1))
    return 0.0


dur = {}
for r in csv.DictReader(open(d + '/kernel_stats.csv')):
    m = re.search(r'k_\w+', r['Name'])
    if m and int(r['Calls']) < 10:      # table builders of the first call, not part of a step
        continue
    if m and m.group(0) not in dur:
        dur[m.group(0)] = float(r['AverageNs']) * 1e-9
print('# Per-kernel resource use (one lane: `bench.py --streams 1`, 100 rows x 35 lambda x 512^2)\n')
print('Source: `scripts/prof_table.sh` -- rocprofv3 kernel trace (durations) and separate PMC passes.')
print('VALU = SQ_INSTS_VALU x 2 issue cycles / (1024 SIMDs x kernel cycles at 2.4 GHz), a lower bound: fp64 '
      'and transcendental instructions take 4; LDS = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles); '
      'HBM = (FETCH_SIZE + WRITE_SIZE) / duration, uncorrected.\n')
print('MFMA = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles).\n')
print('| kernel | avg us | VALU issue | MFMA busy | LDS array busy | of which bank conflicts | HBM GB/s |')
print('|---|---|---|---|---|---|---|')
util = {}
for k in ('k_otf_mfma2', 'k_mf_finish', 'k_mf_prep', 'k_otf_mfma1', 'k_otf_mfma', 'k_otf_r16', 'k_otf_rowfft', 'k_fit', 'k_conv_fft', 'k_colpass_m',
          'k_colpass', 'k_dphi_series', 'k_patch_rows', 'k_patch_gen', 'k_dmin16',
          'k_psd_rowfft', 'k_colfft_dphi', 'k_dmin', 'k_vkeep', 'k_task_order', 'k_khat', 'k_stamp_sum',
          ):
    if k not in dur:
        continue
    t = dur[k]
    cyc = t * 2.4e9
    valu = get(k, 'SQ_INSTS_VALU') * 2 / (1024 * cyc)
    lds = get(k, 'SQ_LDS_IDX_ACTIVE') / (256 * cyc)
    conf = get(k, 'SQ_LDS_BANK_CONFLICT') / max(get(k, 'SQ_LDS_IDX_ACTIVE'), 1)
    hbm = (get(k, 'FETCH_SIZE') + get(k, 'WRITE_SIZE')) * 1024 / t / 1e9
    mfma = get(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * cyc)
    print('| `%s` | %.1f | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.0f |' % (k, t * 1e6, valu * 100, mfma * 100,
                                                                         lds * 100, conf * 100, hbm))
    util[k] = {'avg_us': round(t * 1e6, 1), 'valu_issue': round(valu, 3), 'mfma_busy': round(mfma, 3),
               'lds_array_busy': round(lds, 3),
               'lds_conflict_share': round(conf, 3), 'hbm_GBps': round(hbm, 1)}
    wc = get(k, 'SQ_WAVE_CYCLES')
    if wc:      # where the wave cycles go (disjoint): issuing / issue stall (of which LDS) / waitcnt
        util[k].update(wave_active=round(get(k, 'SQ_ACTIVE_INST_ANY') / wc, 3),
                       wave_issue_stall=round(get(k, 'SQ_WAIT_INST_ANY') / wc, 3),
                       wave_issue_stall_lds=round(get(k, 'SQ_WAIT_INST_LDS') / wc, 3),
                       wave_waitcnt=round(get(k, 'SQ_WAIT_ANY') / wc, 3))
    if k == 'k_otf_r16':     # the per-wavelength kernel under the name bench.py looks up
        util['k_otf_rowfft'] = dict(util[k], kernel_symbol='k_otf_r16')
if len(sys.argv) > 2:
    json.dump(util, open(sys.argv[2], 'w'), indent=1)
