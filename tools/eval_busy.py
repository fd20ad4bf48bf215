"""GPU busy time vs wall time of the 512^3 octree evaluation: union of kernel intervals in a rocprofv3 kernel trace of
tools/recon_host_breakdown.py between the first k_oct_begin and the last k_oct_fill of the LAST evaluation.  usage: eval_busy.py <trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
def short(n): return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
begins = [i for i, r in enumerate(rows) if short(r['Kernel_Name']).startswith('k_oct_begin')]
lo = begins[-1]
hi = max(i for i, r in enumerate(rows) if short(r['Kernel_Name']).startswith('k_oct_scatter_values') and i > lo)
seq = rows[lo:hi + 1]
t0, t1 = int(seq[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in seq)
busy, cur_s, cur_e = 0, None, None
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps = []
prev = int(seq[0]['End_Timestamp'])
for r in seq[1:]:
    s = int(r['Start_Timestamp'])
    if s > prev: gaps.append((s - prev, short(r['Kernel_Name'])))
    prev = max(prev, int(r['End_Timestamp']))
print('span %.2f ms, GPU busy %.2f ms (%.1f %%), %d launches' % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(seq)))
gaps.sort(reverse=True)
print('largest gaps (us, before kernel):', [(round(g / 1e3, 1), k) for g, k in gaps[:12]])
print('sum of gaps > 20 us: %.2f ms in %d gaps' % (sum(g for g, _ in gaps if g > 20000) / 1e6, sum(1 for g, _ in gaps if g > 20000)))
