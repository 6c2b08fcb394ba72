"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel (per dispatch)."""
import csv, glob, re, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_\w+(<[^>]*>)?)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print('   %-28s n=%4d mean=%.4g' % (c, len(v), sum(v) / len(v)))
