"""How the depth walk behaves when fewer layer-0 channels change sign along a column: the bench's name-seeded weights make
nearly every one of the 1024 channels flip inside a 256-row column (the maximum: a channel flips at most once, its
pre-activation is monotone in depth), trained weights need not.  The depth weight of coarse layer 0 is scaled by s: the
sign changes per column fall with s, the field changes, the work of every other kernel does not.
usage: python tools/walk_sensitivity.py [res]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, _lib, dist as pdist
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
oc = config.coarse_opt(); of = config.fine_opt(oc)
spec = {}
synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
synth.mlp_keys('mlp', of.mlp_dim, of.mlp_res_layers, of.mlp_norm, spec)
g = torch.Generator().manual_seed(4242)
fc, ff = torch.randn(1, 256, 128, 128, generator=g), torch.randn(1, 16, 512, 512, generator=g)
calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
for scale in (1.0, 0.5, 0.25, 0.1, 0.0):
    sd = synth.name_seeded_state_dict(spec, seed=0, mlp_std=0.16)
    sd['netG.mlp.filters.0.weight'][:, 256] *= scale
    netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda'); netMR = nets.PIFuMRNet(of, netG); netG.eval()
    netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}; netG._push_weights()
    netMR._sd = {k: t.cuda() for k, t in sd.items() if not k.startswith('netG.')}; netMR._push_weights()
    netG.set_features(fc); netMR.set_features(ff)
    _lib.call('pifu_reserve', netMR._ctx.h, 2097152)
    res = {}
    for flags, tag in ((_lib.Q_FULL_COARSE, 'walk'), (_lib.Q_FULL_COARSE | _lib.Q_NO_DEPTH_WALK, 'gemm')):
        f = pdist.eval_lattice_sharded(netMR, calib, (R, R, R), 5000, flags)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f = pdist.eval_lattice_sharded(netMR, calib, (R, R, R), 5000, flags)
        torch.cuda.synchronize(); res[tag] = (R ** 3 / (time.perf_counter() - t0) / 1e6, f)
    print('depth weight x %.2f: walk %.2f Mq/s, layer-1 GEMM instead %.2f Mq/s, max |diff| %.2e' % (
        scale, res['walk'][0], res['gemm'][0], float((res['walk'][1] - res['gemm'][1]).abs().max())), flush=True)
    del netG, netMR
