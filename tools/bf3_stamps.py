import ctypes as C, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools/lab'); import lab; lab.use()      # lab build: the PIFU_* switches live there
os.environ['PIFU_PRECISION'] = '1'; os.environ['PIFU_DEBUG_STAMPS'] = '1'; os.environ['PIFU_COLUMN_MODE'] = '0'
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
for _ in range(2):
    pdist.eval_lattice_sharded(netG, calib, (64, 64, 64), 5000, _lib.Q_COARSE_ONLY)
torch.cuda.synchronize()
raw = np.zeros(64 * 40 * 4 + 16384 * 16, dtype=np.uint64)
lib = _lib.load(); lib.pifu_debug_read.restype = C.c_int; lib.pifu_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
print('rc', lib.pifu_debug_read(netMR._ctx.h, raw.ctypes.data_as(C.c_void_p), raw.nbytes))
buf = raw[:64 * 40 * 4].reshape(64, 40, 4).astype(np.int64)
for b in (0, 9, 33):
    t = buf[b]; n = int((t[:, 0] > 0).sum())
    print('block', b, 'steps', n)
    print('  gload :', (t[:n, 1] - t[:n, 0])[:14].tolist())
    print('  mfma  :', (t[:n, 2] - t[:n, 1])[:14].tolist())
    print('  lstore:', (t[:n, 3] - t[:n, 2])[:14].tolist())
    print('  period:', np.diff(t[:n, 0])[:14].tolist())
