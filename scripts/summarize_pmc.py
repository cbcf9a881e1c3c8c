#!/usr/bin/env python3
"""Average every collected counter per kernel (smd::* kernels only) from rocprofv3 counter_collection CSVs."""
import csv, glob, sys, collections, os
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(root, 'raw*', '**', '*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'smd::' not in k: continue
        a = acc[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
for k, cs in acc.items():
    print(k[:110])
    for c, (tot, n) in cs.items(): print(f'    {c:34s} avg/dispatch {tot/n:16.1f}   (n={n})')
if not acc:
    print('no counters found; files:', glob.glob(os.path.join(root, '**', '*'), recursive=True)[:20])
    for f in glob.glob(os.path.join(root, 'run*.log')): print(open(f).read()[-1500:])
