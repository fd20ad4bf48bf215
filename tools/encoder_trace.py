"""Launch-ordered view of ONE filter_global + filter_local pass from a rocprofv3 kernel trace of tools/encoder_only.py
(the last of its three passes): consecutive launches of the same (kernel, grid) are merged, so the conv layers can be read
off in network order.  usage: encoder_trace.py <kernel_trace.csv> [min_us]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0


def short(n):
    return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]


# the passes start with k_pack_input of the 512^2 image (filter_global): take everything from the last but one occurrence
starts = [i for i, r in enumerate(rows) if short(r['Kernel_Name']).startswith('k_pack_input')]
lo = starts[-2] if len(starts) >= 2 else 0          # each pass packs twice (coarse, fine)
seq = rows[lo:]
total = 0.0
agg = []
for r in seq:
    name = short(r['Kernel_Name'])
    blocks = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    total += us
    if agg and agg[-1][0] == name and agg[-1][1] == blocks:
        agg[-1][2] += 1
        agg[-1][3] += us
    else:
        agg.append([name, blocks, 1, us])
for name, blocks, n, us in agg:
    if us >= min_us:
        print('%-28s blocks %7d  x%-3d %9.1f us' % (name, blocks, n, us))
span = (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3
print('kernel time %.2f ms, span %.2f ms, %d launches' % (total / 1e3, span / 1e3, len(seq)))
by = {}
for name, blocks, n, us in agg:
    by[name] = by.get(name, 0.0) + us
for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
    print('  %-28s %8.2f ms' % (k, v / 1e3))
