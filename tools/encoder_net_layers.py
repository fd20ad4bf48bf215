"""Launch-ordered view of ONE normal net (netB = the library's lane-1 stream) of the last encoder pass in a rocprofv3 kernel
trace of tools/encoder_only.py: consecutive launches of the same (kernel, grid) merged.  usage: encoder_net_layers.py <trace.csv> [queue]"""
import csv, sys
from collections import Counter
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
def short(n): return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
packs = [i for i, r in enumerate(rows) if short(r['Kernel_Name']).startswith('k_pack_input')]
seq = rows[packs[-3]:]
qs = Counter(r['Queue_Id'] for r in seq)
q = sys.argv[2] if len(sys.argv) > 2 else sorted(qs, key=lambda k: qs[k])[0]      # the queue with the fewest launches = netB
agg = []
tot = 0.0
for r in seq:
    if r['Queue_Id'] != q: continue
    name = short(r['Kernel_Name']); blocks = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += us
    if agg and agg[-1][0] == name and agg[-1][1] == blocks: agg[-1][2] += 1; agg[-1][3] += us
    else: agg.append([name, blocks, 1, us])
print('queue', q, 'launches', qs[q], 'kernel time %.2f ms' % (tot / 1e3))
for name, blocks, n, us in agg:
    print('%-24s blocks %6d x%-3d %8.1f us' % (name, blocks, n, us))
