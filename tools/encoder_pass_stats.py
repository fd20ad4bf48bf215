"""Per-kernel statistics of ONE steady-state encoder pass (the last `reconstruction._Subject` pass in a rocprofv3 kernel trace
of tools/encoder_only.py): calls, average and total duration, and every runtime copy / fill kernel of that pass by name -
the whole-run `--stats` table mixes in the first pass, whose ~2 000 copyBuffer calls are the one-time weight uploads.
usage: encoder_pass_stats.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))


def short(n):
    return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]


packs = [i for i, r in enumerate(rows) if short(r['Kernel_Name']).startswith('k_pack_prepared')]
seq = rows[packs[-3]:]                 # a pass packs three times: normal-net input, coarse hourglass input, fine input
agg = defaultdict(lambda: [0, 0.0])
for r in seq:
    a = agg[short(r['Kernel_Name'])]
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
total = sum(v[1] for v in agg.values())
span = (max(int(r['End_Timestamp']) for r in seq) - int(seq[0]['Start_Timestamp'])) / 1e3
print('one pass: %d launches, kernel time %.2f ms, first start -> last end %.2f ms' % (len(seq), total / 1e3, span / 1e3))
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-40s calls %5d avg_us %8.1f total_ms %8.2f %6.2f%%' % (k[:40], n, us / n, us / 1e3, 100 * us / total))
copies = {k: v for k, v in agg.items() if 'copy' in k.lower() or 'fill' in k.lower()}
print('runtime copy / fill kernels in the pass:', {k: v[0] for k, v in copies.items()} or 'none')
