"""Per-layer rate of the MLP GEMM on the headline launch mix, from the rocprofv3 kernel trace of tools/profile_gpu.sh
(gpurun_out/prof/trace): the big launches of a batch come in the order coarse L2, coarse L3, fine L0, fine L1, fine L2, so
launches of equal grid size are told apart by their position.  usage: python tools/layer_fracs.py [kernel_trace.csv]"""
import collections, csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else glob.glob('gpurun_out/prof/trace/*/*_kernel_trace.csv')[0]
rows = sorted((r for r in csv.DictReader(open(f)) if 'k_mlp_gemm_tri' in r['Kernel_Name']), key=lambda r: int(r['Start_Timestamp']))
sizes = collections.Counter(int(r['Grid_Size_X']) // 256 for r in rows)
ntm = max(sizes) // 4                       # row tiles of a full batch (the N = 512 launch has 4 column tiles)
valid = ntm // 40 * 5000                    # 40 tiles per 5000-point chunk
seq = {2 * ntm: [('coarse L2', 256, 512), ('fine L1', 256, 768)], ntm: [('coarse L3', 128, 256), ('fine L2', 128, 512)],
       4 * ntm: [('fine L0', 512, 256)]}
cnt, agg = collections.Counter(), collections.defaultdict(list)
for r in rows:
    b = int(r['Grid_Size_X']) // 256
    if b in seq:
        name, N, K = seq[b][cnt[b] % len(seq[b])]
        cnt[b] += 1
        agg[(name, N, K)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('| layer | N | K | K steps | avg us | TFLOP/s | of 157.3 |')
print('|---|---|---|---|---|---|---|')
for (name, N, K), v in sorted(agg.items(), key=lambda kv: (kv[0][2], kv[0][1])):
    us = sum(v) / len(v)
    tf = 2 * valid * N * K / (us * 1e-6) / 1e12
    print('| %s | %d | %d | %d | %.0f | %.1f | %.3f |' % (name, N, K, K // 32, us, tf, tf / 157.3))
