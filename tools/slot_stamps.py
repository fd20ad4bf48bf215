"""Block life-cycle stamps of k_mlp_gemm_buf2 (EXPERIMENTS build, PIFU_GEMM_VARIANT=6): prologue, main loop, epilogue,
and the dispatch gap of a CU slot.  All in shader-clock cycles (s_memtime, per-XCD counters: only same-CU differences)."""
import ctypes as C, os, sys, collections
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/lab'); import lab; lab.use()      # lab build: the PIFU_* switches live there
os.environ['PIFU_GEMM_VARIANT'] = '6'; os.environ['PIFU_DEBUG_STAMPS'] = '1'
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, _lib, dist as pdist
oc = config.coarse_opt(); of = config.fine_opt(oc)
spec = {}
synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
sd = synth.name_seeded_state_dict(spec, seed=0, mlp_std=0.16)
g = torch.Generator().manual_seed(4242)
netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda'); netG.eval()
netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}; netG._push_weights()
netG.set_features(torch.randn(1, 256, 128, 128, generator=g))
calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
_lib.call('pifu_reserve', netG._ctx.h, 262144)
for _ in range(2):
    pdist.eval_lattice_sharded(netG, calib, (51, 50, 100), 5000, _lib.Q_COARSE_ONLY)
torch.cuda.synchronize()
raw = np.zeros(64 * 40 * 4 + 16384 * 16, dtype=np.uint64)
lib = _lib.load(); lib.pifu_debug_read.restype = C.c_int; lib.pifu_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
print('rc', lib.pifu_debug_read(netG._ctx.h, raw.ctypes.data_as(C.c_void_p), raw.nbytes))
cen = raw[64 * 40 * 4:].reshape(16384, 16).astype(np.int64)
nb = int((cen[:, 0] > 0).sum()); cen = cen[:nb]; print('blocks', nb)
tb, tl, hw, xcc, tf, te = cen[:, 0], cen[:, 1], cen[:, 2], cen[:, 3] & 15, cen[:, 4], cen[:, 5]
print('prologue (begin -> first MFMA): median', int(np.median(tf - tb)), 'p90', int(np.percentile(tf - tb, 90)))
print('main loop: median', int(np.median(tl - tf)))
print('epilogue (loop end -> stores acknowledged): median', int(np.median(te - tl)), 'p90', int(np.percentile(te - tl, 90)))
key = (xcc << 16) | ((hw >> 8) & 255)
per = collections.defaultdict(list)
for b in range(nb): per[int(key[b])].append((int(tb[b]), int(te[b]), int(tf[b]), int(tl[b])))
gaps = []; both_out = []
for k, iv in per.items():
    iv.sort()
    slots = [[], []]
    for it in iv:
        for sl in slots:
            if not sl or sl[-1][1] <= it[0]:
                sl.append(it); break
    for sl in slots:
        for p, n in zip(sl[:-1], sl[1:]): gaps.append(n[0] - p[1])
    # time during which NO block of the CU is inside its main loop
    ev = sorted([(it[2], 1) for it in iv] + [(it[3], -1) for it in iv])
    cur = 0; last = ev[0][0]; w = collections.Counter()
    for t, d in ev:
        w[cur] += t - last; last = t; cur += d
    tot = sum(w.values()); both_out.append([w[0] / tot, w[1] / tot, w[2] / tot])
print('dispatch gap (stores acknowledged -> next block begins in the slot): median', int(np.median(gaps)), 'p10', int(np.percentile(gaps, 10)), 'p90', int(np.percentile(gaps, 90)))
print('share of time with 0 / 1 / 2 blocks inside their main loop (mean over CUs):', np.round(np.mean(both_out, 0), 3).tolist())
