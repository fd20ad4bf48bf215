"""Per (kernel, grid) histogram of ONE filter_global + filter_local pass from a rocprofv3 kernel trace of tools/encoder_only.py
(the last of its three passes).  usage: encoder_hist.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))


def short(n):
    return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]


# a pass = [k_pack_input (normal nets' input) ... ]: passes start where a k_pack_input follows a k_nhwc_to_nchw / the start
starts = [i for i, r in enumerate(rows) if short(r['Kernel_Name']).startswith('k_pack_input')]
per_pass = len(starts) // 3
lo = starts[-per_pass] if per_pass else 0
seq = rows[lo:]
agg = {}
total = 0.0
for r in seq:
    name = short(r['Kernel_Name'])
    blocks = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    total += us
    k = (name, blocks)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += us
print('pass: %d launches, kernel time %.2f ms, span %.2f ms' % (len(seq), total / 1e3, (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e6))
for (name, blocks), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%-26s blocks %7d  x%-4d %9.1f us total  %8.1f us each' % (name, blocks, n, us, us / n))
