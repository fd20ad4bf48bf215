"""Depth walk on/off on the same lattice: throughput and max difference of the occupancy field.
usage: python tools/ab_walk.py [res]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, _lib, dist as pdist
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
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
out = {}
for flags, tag in ((_lib.Q_FULL_COARSE, 'walk'), (_lib.Q_FULL_COARSE | _lib.Q_NO_DEPTH_WALK, 'gemm'), (_lib.Q_FULL_COARSE | _lib.Q_NO_COLUMN_SHARING, 'per-point')):
    f = pdist.eval_lattice_sharded(netMR, calib, (R, R, R), 5000, flags)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = pdist.eval_lattice_sharded(netMR, calib, (R, R, R), 5000, flags)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[tag] = f
    print(tag, '%.2f Mq/s' % (R ** 3 / dt / 1e6), 'mean %.4f' % float(f.mean()), flush=True)
print('max |walk - gemm| %.3e   max |walk - per-point| %.3e   max |gemm - per-point| %.3e' % (
    float((out['walk'] - out['gemm']).abs().max()), float((out['walk'] - out['per-point']).abs().max()),
    float((out['gemm'] - out['per-point']).abs().max())))
f2 = pdist.eval_lattice_sharded(netMR, calib, (R, R, R), 5000, _lib.Q_FULL_COARSE)
print('walk rerun bit-identical:', bool(torch.equal(f2, out['walk'])))
