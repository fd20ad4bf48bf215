"""Per-kernel time of one filter_global + filter_local pass from a rocprofv3 kernel trace (tools: rocprofv3 --kernel-trace
-- python tools/genmesh_profile.py).  usage: encoder_times.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    if n.startswith(('k_mlp', 'k_norm_f', 'k_last', 'k_expand', 'k_sample', 'k_oct', 'k_mc', 'k_scan', 'k_normals', 'k_phi')) or 'rocclr' in n or 'at::' in n or 'elementwise' in n:
        continue
    key = (n, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    tot[key] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; cnt[key] += 1
for k, v in tot.most_common(16):
    print(k, cnt[k], 'total ms %.2f' % (v / 1e3), 'avg us %.0f' % (v / cnt[k]))
print('sum ms %.2f' % (sum(tot.values()) / 1e3))
