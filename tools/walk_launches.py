#!/usr/bin/env python3
"""Per-launch view of the depth walk in a rocprofv3 kernel trace (one octree evaluation's launches, in issue order): grid (=
column segments), duration, microseconds per segment - for the body-like subject's short segments (round 6).
usage: walk_launches.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
names = ('k_walk', 'k_seg_prepare', 'k_seg_stats0', 'k_expand', 'k_oct_fill', 'k_oct_cells', 'k_mlp_gemm_tri')
for r in rows[-400:]:
    n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    if not n.startswith(names):
        continue
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gx, gy = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1)
    print('%-28s blocks %8d x %d  %9.1f us  %7.3f us/block' % (n[:28], gx, gy, us, us / max(gx * gy, 1)))
