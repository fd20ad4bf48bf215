#!/usr/bin/env python3
"""Per-layer GEMM durations from a rocprofv3 --kernel-trace CSV: launches are grouped by (kernel, grid size), which
identifies the layer in the column-sharing pipeline.  usage: layer_times.py <kernel_trace.csv>"""
import collections
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'gemm' not in n:
        continue
    short = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    agg[(short, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    # bimodal groups (two layers with the same grid) are split at the midpoint of min/max
    mid = (v.min() + v.max()) / 2
    lo, hi = v[v <= mid], v[v > mid]
    if len(lo) > 10 and len(hi) > 10 and v.max() > 1.5 * v.min():
        print(k, 'n', len(v), 'modes: median %.0f us (n %d) | median %.0f us (n %d)' % (np.median(lo), len(lo), np.median(hi), len(hi)))
    else:
        print(k, 'n', len(v), 'median %.0f us' % np.median(v))
