import ctypes as C, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/lab'); import lab; lab.use()      # lab build: the PIFU_* switches live there
os.environ.setdefault('PIFU_GEMM_VARIANT', '9'); os.environ['PIFU_DEBUG_STAMPS'] = '1'
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, _lib, dist as pdist
oc = config.coarse_opt(); of = config.fine_opt(oc)
spec = {}
synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
synth.mlp_keys('mlp', of.mlp_dim, of.mlp_res_layers, of.mlp_norm, spec)
sd = synth.name_seeded_state_dict(spec, seed=0, mlp_std=0.16)
g = torch.Generator().manual_seed(4242)
netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda'); netMR = nets.PIFuMRNet(of, netG); netG.eval()
netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}; netG._push_weights()
netMR._sd = {k: t.cuda() for k, t in sd.items() if not k.startswith('netG.')}; netMR._push_weights()
netG.set_features(torch.randn(1, 256, 128, 128, generator=g)); netMR.set_features(torch.randn(1, 16, 512, 512, generator=g))
calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
_lib.call('pifu_reserve', netMR._ctx.h, 262144)
# coarse only up to layer 1 would be ideal; we simply run the coarse-only path: the LAST gemm launch (coarse L3, K=544)
# overwrites the stamps, so use a 2-layer trick: evaluate coarse only and read stamps of the last GEMM launched.
for _ in range(2):
    pdist.eval_lattice_sharded(netG, calib, (51, 50, 100), 5000, _lib.Q_COARSE_ONLY)   # exactly one batch of 51 chunks
torch.cuda.synchronize()
raw = np.zeros(64 * 40 * 4 + 16384 * 16, dtype=np.uint64)
buf = raw[:64 * 40 * 4].reshape(64, 40, 4)
lib = _lib.load(); lib.pifu_debug_read.restype = C.c_int; lib.pifu_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
print('rc', lib.pifu_debug_read(netMR._ctx.h, raw.ctypes.data_as(C.c_void_p), raw.nbytes))
for b in (0, 1, 8, 9, 17, 33):
    t = buf[b].astype(np.int64)
    n = int((t[:, 0] > 0).sum())
    if n < 2: continue
    hw = int(buf[b, 0, 3])
    print(f'block {b}: steps {n} hw_id=0x{hw:08x} simd={(hw>>4)&3} cu={(hw>>8)&15} wave={hw&15}')
    mf = (t[:n, 1] - t[:n, 0]); st = (t[:n, 2] - t[:n, 1]); gap = (t[1:n, 0] - t[:n-1, 2])
    print('   mfma phase cyc:', mf[:12].tolist()); print('   staging  cyc:', st[:12].tolist()); print('   barrier+next-loads gap:', gap[:12].tolist())
    print('   step period:', (t[1:n,0]-t[:n-1,0])[:12].tolist())

cen = raw[64 * 40 * 4:].reshape(16384, 4, 4).astype(np.int64)
nb = int((cen[:, 0, 0] > 0).sum()); print('blocks recorded', nb)
cen = cen[:nb]
hw = cen[:, :, 2]; xcc = cen[:, :, 3] & 0xF
cu_key = (xcc[:, 0] << 16) | (((hw[:, 0] >> 13) & 7) << 8) | (((hw[:, 0] >> 12) & 1) << 4) | ((hw[:, 0] >> 8) & 15)
print('distinct CUs', len(set(cu_key.tolist())))
simds = (hw >> 4) & 3
print('SIMD ids of the 4 waves, first 6 blocks:', simds[:6].tolist())
import collections
per = collections.defaultdict(list)
for b in range(nb): per[int(cu_key[b])].append((int(cen[b, 0, 0]), int(cen[b, :, 1].max())))
conc = []
for k, iv in per.items():
    ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    cur = 0; last = ev[0][0]; w = collections.Counter()
    for t, d in ev:
        w[cur] += t - last; last = t; cur += d
    tot = sum(w.values()); conc.append({c: round(v / tot, 3) for c, v in sorted(w.items())})
print('time share by number of resident blocks, first 5 CUs:', conc[:5])
dur = (cen[:, :, 1].max(1) - cen[:, 0, 0]); print('block duration cyc: median', int(np.median(dur)), 'min', int(dur.min()), 'max', int(dur.max()))

# ---- per-CU timelines: gaps between consecutive blocks of a slot, phase offset of the two resident blocks
gaps = []; phases = []; starts0 = []
for k, iv in per.items():
    iv = sorted(iv)
    # assign to 2 slots greedily
    slots = [[], []]
    for s, e in iv:
        for sl in slots:
            if not sl or sl[-1][1] <= s:
                sl.append((s, e)); break
    for sl in slots:
        for (s0, e0), (s1, e1) in zip(sl[:-1], sl[1:]):
            gaps.append(s1 - e0)
    if slots[0] and slots[1]:
        for (s0, e0) in slots[0][1:4]:
            # nearest block of the other slot that overlaps
            for (s1, e1) in slots[1]:
                if s1 < e0 and e1 > s0:
                    phases.append(((s1 - s0) % (e0 - s0)) / (e0 - s0)); break
gaps = np.array(gaps); phases = np.array(phases)
print('slot gap (end of a block -> begin of the next on the same slot) cyc: median', int(np.median(gaps)), 'p10', int(np.percentile(gaps, 10)), 'p90', int(np.percentile(gaps, 90)))
print('phase offset of the co-resident block (fraction of a tile): hist', np.histogram(phases, bins=10, range=(0, 1))[0].tolist())
t0 = cen[:, 0, 0].min()
print('first 16 block begins (cyc since first):', sorted((cen[:, 0, 0] - t0).tolist())[:16])
print('begin of blocks 500..530:', (cen[500:530, 0, 0] - t0).tolist())
print('kernel span cyc', int(cen[:, :, 1].max() - t0), 'sum of block durations / (span * 512)', float(dur.sum() / ((cen[:, :, 1].max() - t0) * 512.0)))
