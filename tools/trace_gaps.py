#!/usr/bin/env python3
"""Digest of a `rocprofv3 --kernel-trace` csv: per kernel name the launch count, duration (mean / median / p10 / p90) and
the idle gap between a launch's start and the end of the launch before it on the device.
usage: trace_gaps.py <dir or *_kernel_trace.csv> [name substring]"""
import csv, pathlib, statistics, sys
from collections import defaultdict

root = pathlib.Path(sys.argv[1])
files = [root] if root.is_file() else sorted(root.rglob('*kernel_trace.csv'))
want = sys.argv[2] if len(sys.argv) > 2 else ''
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
by = defaultdict(lambda: ([], []))
prev_end = None
for s, e, name in rows:
    d, g = by[name]
    d.append((e - s) / 1e3)
    if prev_end is not None:
        g.append((s - prev_end) / 1e3)
    prev_end = e
for name, (d, g) in sorted(by.items(), key=lambda kv: -sum(kv[1][0])):
    if want and want not in name:
        continue
    d2 = sorted(d); g2 = sorted(g) or [0.0]
    q = lambda a, p: a[min(len(a) - 1, int(p * len(a)))]
    print(f'{name[:70]:70s} n={len(d):6d}  duration us mean {statistics.fmean(d):8.2f} median {q(d2, .5):8.2f} p10 {q(d2, .1):8.2f} p90 {q(d2, .9):8.2f}'
          f'   gap before us median {q(g2, .5):6.2f} p10 {q(g2, .1):6.2f} p90 {q(g2, .9):6.2f}')
