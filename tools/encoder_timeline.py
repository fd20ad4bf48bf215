"""Per-stream timeline of the LAST encoder pass in a rocprofv3 kernel trace of tools/encoder_only.py: for every queue the
first start, last end, summed kernel time and launch count, and the pass's overall span; then the same split in 1 ms bins
(kernel time per queue and bin), which shows what overlaps with what.  usage: encoder_timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))


def short(n):
    return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]


packs = [i for i, r in enumerate(rows) if short(r['Kernel_Name']).startswith('k_pack_prepared')]
# a pass packs three times (normal-net input, coarse hourglass input, fine input)
lo = packs[-3] if len(packs) >= 3 else 0
seq = rows[lo:]
t0 = int(seq[0]['Start_Timestamp'])
qkey = 'Queue_Id' if 'Queue_Id' in seq[0] else ('Stream_Id' if 'Stream_Id' in seq[0] else None)
print('columns:', ', '.join(seq[0].keys()))
per = defaultdict(lambda: [None, 0, 0.0, 0])
for r in seq:
    q = r.get(qkey, '?')
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    p = per[q]
    p[0] = s if p[0] is None else min(p[0], s)
    p[1] = max(p[1], e)
    p[2] += (e - s)
    p[3] += 1
for q, (s, e, busy, n) in sorted(per.items(), key=lambda kv: kv[1][0]):
    print('queue %-6s first %8.2f ms  last %8.2f ms  kernel time %7.2f ms  launches %4d' % (q, s / 1e6, e / 1e6, busy / 1e6, n))
span = max(p[1] for p in per.values())
print('span %.2f ms, summed kernel time %.2f ms' % (span / 1e6, sum(p[2] for p in per.values()) / 1e6))
nb = int(span / 1e6) + 1
bins = defaultdict(lambda: [0.0] * nb)
for r in seq:
    q = r.get(qkey, '?')
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    b = int(s / 1e6)
    while s < e and b < nb:
        hi = min(e, (b + 1) * 1000000)
        bins[q][b] += (hi - s) / 1e6
        s = hi
        b += 1
for q in sorted(bins):
    print('queue %-6s' % q, ' '.join('%4.2f' % v for v in bins[q]))
# the biggest kernels of the pass by queue
top = defaultdict(float)
for r in seq:
    top[(r.get(qkey, '?'), short(r['Kernel_Name']))] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
for (q, k), v in sorted(top.items(), key=lambda kv: -kv[1])[:14]:
    print('  queue %-6s %-24s %7.2f ms' % (q, k, v))
