#!/usr/bin/env python3
"""Digest rocprofv3 CSV output of tools/profile.sh: per-kernel mean duration from
the kernel trace and per-dispatch mean of every PMC counter."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {'kernels': {}, 'counters': {}}


def short(name):
    return name.split('(')[0].replace('void ', '').strip()


for path in glob.glob(os.path.join(root, 'trace', '**', '*kernel_trace.csv'), recursive=True):
    dur = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            dur[short(row['Kernel_Name'])].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
    for k, v in dur.items():
        v.sort()
        out['kernels'][k] = {'calls': len(v), 'mean_us': sum(v) / len(v) / 1e3, 'median_us': v[len(v) // 2] / 1e3,
                             'min_us': v[0] / 1e3, 'max_us': v[-1] / 1e3}
for path in glob.glob(os.path.join(root, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    out['stats_csv'] = open(path).read().splitlines()[:12]

for path in glob.glob(os.path.join(root, '*', '**', '*counter_collection.csv'), recursive=True):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items():
            out['counters'].setdefault(k, {})[c] = {'mean': sum(v) / len(v), 'n': len(v)}

print(json.dumps(out, indent=1, sort_keys=True))
