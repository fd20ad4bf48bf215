#!/usr/bin/env python3
"""The launches of the LAST gen_mesh call in a rocprofv3 kernel trace of tools/genmesh_profile.py: per kernel launches and
summed time, runtime copy / fill kernels included, and how much of the call's span no kernel covers (idle gaps, by size).
A call starts at its first k_pack_prepared launch (the encoders' first kernel) and ends before the next one's.
usage: call_window.py <kernel_trace.csv> [n_calls_back=1]"""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def short(n):
    return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:44]


starts = []
last = -10 ** 18
for i, r in enumerate(rows):
    if 'k_pack_prepared' in r['Kernel_Name']:
        t = int(r['Start_Timestamp'])
        if t - last > 20_000_000:          # a new call: the three packs of one filter pass sit within a few ms
            starts.append(i)
        last = t
if len(starts) < back:
    sys.exit('no gen_mesh call found in the trace')
lo = starts[-back]
hi = starts[-back + 1] if back > 1 else len(rows)
win = rows[lo:hi]
# the tool runs one more octree / marching-cubes pass after its last call: cut the window at the first gap > 30 ms
cut = len(win)
for i in range(1, len(win)):
    if int(win[i]['Start_Timestamp']) - int(win[i - 1]['End_Timestamp']) > 30_000_000:
        cut = i
        break
win = win[:cut]
t0, t1 = int(win[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in win)
per = defaultdict(lambda: [0, 0])
for r in win:
    p = per[short(r['Kernel_Name'])]
    p[0] += 1
    p[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print(f'last call: {len(win)} launches, span {(t1 - t0) / 1e6:.2f} ms, summed kernel time {sum(p[1] for p in per.values()) / 1e6:.2f} ms')
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f'{n:46s} calls {c:5d} total_ms {t / 1e6:8.3f} avg_us {t / c / 1e3:8.1f}')
# uncovered time: sweep over the union of the kernel intervals
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in win)
gaps = []
end = iv[0][1]
for s, e in iv[1:]:
    if s > end:
        gaps.append((s - end, end - t0))
    end = max(end, e)
tot = sum(g for g, _ in gaps)
print(f'no kernel running: {tot / 1e6:.2f} ms in {len(gaps)} gaps; '
      + ', '.join(f'>{b} us: {sum(g for g, _ in gaps if g > b * 1e3) / 1e6:.2f} ms in {sum(1 for g, _ in gaps if g > b * 1e3)}' for b in (20, 100, 500)))
for g, at in sorted(gaps, reverse=True)[:12]:
    print(f'  gap {g / 1e3:8.1f} us at {at / 1e6:7.2f} ms')
