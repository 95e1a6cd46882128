#!/usr/bin/env python3
"""pmc_summ.py DIR: average per-launch counter values of the k_fused* kernels per variant label."""
import csv, collections, glob, os, sys
d = sys.argv[1]
res = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, '*', '*', '*counter_collection.csv'))):
    label = f[len(d):].strip('/').split('/')[0].rsplit('_', 1)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_fused' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        res[k][label] = sum(v) / len(v)
labels = sorted({l for v in res.values() for l in v})
print(f"{'counter':34s}" + ''.join(f"{l:>14s}" for l in labels))
for k in sorted(res):
    print(f"{k:34s}" + ''.join(f"{res[k].get(l, float('nan')):14.4g}" for l in labels))
