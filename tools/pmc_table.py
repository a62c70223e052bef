#!/usr/bin/env python3
"""Per-kernel averages of every counter in rocprofv3 --pmc output directories: pmc_table.py DIR [DIR ...] [--grep SUBSTR]."""
import csv, glob, os, sys
from collections import defaultdict
dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
pat = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--grep=')), 'nerfail')
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                acc[r['Kernel_Name'].split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print('    %-34s n=%-4d avg=%-16.1f max=%.1f' % (c, len(v), sum(v) / len(v), max(v)))
