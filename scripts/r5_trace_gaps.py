"""Per hardware queue of a two-lane kernel trace (rocprofv3 --kernel-trace): kernel durations, the gap in front of
every kernel (time its queue stood still), and what runs beside k_otf_mfma2.
usage: python scripts/r5_trace_gaps.py <dir with *kernel_trace.csv>"""
import collections, csv, glob, re, sys
import numpy as np
fs = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=lambda f: -len(open(f).read()))
rows = list(csv.DictReader(open(fs[0])))
short = lambda n: (re.search(r'(k_\w+)', n) or [None, n[:20]])[1]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], short(r['Kernel_Name'])) for r in rows)
t0, t1 = ev[int(len(ev) * 0.5)][0], ev[-200][0]           # steady state: the second half
ev = [e for e in ev if t0 <= e[0] <= t1]
lanes = [q for q, _ in collections.Counter(e[2] for e in ev).most_common(2)]
for q in lanes:
    es = [e for e in ev if e[2] == q]
    gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in zip(es[:-1], es[1:]):
        gaps[b[3]].append((b[0] - a[1]) / 1e3)
        durs[a[3]].append((a[1] - a[0]) / 1e3)
    print('queue', q, 'kernels', len(es))
    tg = td = 0.0
    for k in sorted(durs, key=lambda k: -np.sum(durs[k])):
        g = gaps.get(k, [0])
        print('  %-16s n %4d  duration mean %6.1f us | gap in front: median %5.1f mean %5.1f p90 %5.1f us' % (
            k, len(durs[k]), np.mean(durs[k]), np.median(g), np.mean(g), np.percentile(g, 90)))
        tg += np.sum(g)
        td += np.sum(durs[k])
    print('  kernels %.1f ms, gaps %.1f ms (%.1f %%), span %.1f ms' % (td / 1e3, tg / 1e3, 100 * tg / (tg + td), (es[-1][1] - es[0][0]) / 1e6))
A, B = ([e for e in ev if e[2] == q] for q in lanes)
def inter(x, y):
    i = j = 0
    out = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        out += max(0, b - a)
        if x[i][1] < y[j][1]: i += 1
        else: j += 1
    return out
span = max(A[-1][1], B[-1][1]) - min(A[0][0], B[0][0])
print('span %.1f ms: lane A busy %.1f %%, lane B busy %.1f %%, both %.1f %%' % (
    span / 1e6, 100 * sum(e[1] - e[0] for e in A) / span, 100 * sum(e[1] - e[0] for e in B) / span, 100 * inter(A, B) / span))
pair = collections.Counter()
for x, y in ((A, B), (B, A)):
    for a in x:
        if a[3] != 'k_otf_mfma2': continue
        for b in y:
            o = min(a[1], b[1]) - max(a[0], b[0])
            if o > 0: pair[b[3]] += o
tm = sum(e[1] - e[0] for e in ev if e[3] == 'k_otf_mfma2')
print('beside k_otf_mfma2 (share of its time):', {k: round(v / tm, 3) for k, v in pair.most_common(8)})
