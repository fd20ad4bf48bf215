"""Tile life-cycle stamps of k_mlp_gemm_pers (build: make EXPERIMENTS=1 CXXFLAGS+=-DPIFU_PERS_STAMPS; PIFU_GEMM_VARIANT=20) on the fine
layer 0 (K 256, N 512): [tile start, main loop end, epilogue end] per tile of the first 512 blocks, shader-clock cycles."""
import ctypes as C, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/lab'); import lab; lab.use()      # lab build: the PIFU_* switches live there
os.environ.setdefault('PIFU_GEMM_VARIANT', '20'); os.environ['PIFU_DEBUG_STAMPS'] = '1'
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
for _ in range(int(os.environ.get('PERS_REPS', '60'))):     # long enough for the clock to reach its sustained level
    pdist.eval_lattice_sharded(netMR, calib, (51, 50, 100), 5000, _lib.Q_FULL_COARSE)     # exactly one batch of 51 chunks; stamps of the last launch survive
torch.cuda.synchronize()
raw = np.zeros(64 * 40 * 4 + 16384 * 16, dtype=np.uint64)
lib = _lib.load(); lib.pifu_debug_read.restype = C.c_int; lib.pifu_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
print('rc', lib.pifu_debug_read(netMR._ctx.h, raw.ctypes.data_as(C.c_void_p), raw.nbytes))
st = raw[:512 * 32 * 4].reshape(512, 32, 4).astype(np.int64)
ok = (st[:, :, 0] > 0) & (st[:, :, 1] > 0) & (st[:, :, 2] > 0)
print('tiles stamped', int(ok.sum()))
loop = (st[:, :, 1] - st[:, :, 0])[ok]; epi = (st[:, :, 2] - st[:, :, 1])[ok]
print('main loop (8 steps): median', int(np.median(loop)), 'p10', int(np.percentile(loop, 10)), 'p90', int(np.percentile(loop, 90)))
print('epilogue: median', int(np.median(epi)), 'p90', int(np.percentile(epi, 90)))
nxt = st[:, 1:, 0] - st[:, :-1, 2]; okn = ok[:, 1:] & ok[:, :-1]
print('epilogue end -> next tile start (acc init): median', int(np.median(nxt[okn])), 'p90', int(np.percentile(nxt[okn], 90)))
tot = st[:, 1:, 0] - st[:, :-1, 0]
print('tile period: median', int(np.median(tot[okn])))
for b in (0, 1, 8, 9):
    print('block', b, 'loop', (st[b, :6, 1] - st[b, :6, 0]).tolist(), 'epi', (st[b, :6, 2] - st[b, :6, 1]).tolist())

# shader clock: s_memtime ticks per s_memrealtime tick (100 MHz) over each block's life
dt = (st[:, :, 0].max(1) - st[:, 0, 0]); dr = (st[:, :, 3].max(1) - st[:, 0, 3])
good = dr > 0
print('s_memtime ticks per 10 ns: median %.3f  => shader clock %.0f MHz (if the real-time counter runs at 100 MHz)' % (float(np.median(dt[good] / dr[good])), float(np.median(dt[good] / dr[good])) * 100))
print('kernel span: %.1f us (real-time counter), %d shader ticks' % (float(st[:, :, 3].max() - st[:, 0, 3][st[:, 0, 3] > 0].min()) / 100.0, int(st[:, :, 2].max() - st[:, 0, 0][st[:, 0, 0] > 0].min())))
# per-block life (first tile start -> last epilogue end) and tiles per block
nt = ok.sum(1)
life = np.array([st[b, nt[b] - 1, 2] - st[b, 0, 0] for b in range(512) if nt[b] > 0])
print('tiles per block: min %d max %d; block life cycles: median %d p10 %d p90 %d max %d' % (nt.min(), nt.max(), np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.max()))
# start skew between blocks of one XCD (same counter): first-start spread over blocks with the same blockIdx % 8
for x in range(2):
    s0 = np.array([st[b, 0, 0] for b in range(x, 512, 8)]); e0 = np.array([st[b, nt[b] - 1, 2] for b in range(x, 512, 8)])
    print('XCD slot %d: first starts span %d cycles, ends span %d cycles, kernel span on this XCD %d cycles' % (x, s0.max() - s0.min(), e0.max() - e0.min(), e0.max() - s0.min()))
lifeb = np.array([st[b, nt[b] - 1, 2] - st[b, 0, 0] if nt[b] > 0 else 0 for b in range(512)])
print('median block life by blockIdx % 8 :', [int(np.median(lifeb[x::8])) for x in range(8)])
print('median block life by (blockIdx // 8) % 4:', [int(np.median(lifeb[np.arange(512) // 8 % 4 == x])) for x in range(4)])
print('tiles by blockIdx % 8:', [int(nt[x::8].sum()) for x in range(8)])
loopb = (st[:, :, 1] - st[:, :, 0]); 
print('median main loop by blockIdx % 8:', [int(np.median(loopb[x::8][ok[x::8]])) for x in range(8)])
