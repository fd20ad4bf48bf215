import sys, time, traceback
sys.path.insert(0, '.')
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, mesh_util, _lib
oc = config.coarse_opt(); of = config.fine_opt(oc)
spec = {}
synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
synth.mlp_keys('mlp', of.mlp_dim, of.mlp_res_layers, of.mlp_norm, spec)
sd = synth.name_seeded_state_dict(spec, seed=0, mlp_std=0.16)
netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda'); netMR = nets.PIFuMRNet(of, netG); netG.eval()
netG._sd = {k[5:]: v.cuda() for k, v in sd.items() if k.startswith('netG.')}; netG._push_weights()
netMR._sd = {k: v.cuda() for k, v in sd.items() if not k.startswith('netG.')}; netMR._push_weights()
g = torch.Generator().manual_seed(4242)
netG.set_features(torch.randn(1, 256, 128, 128, generator=g)); netMR.set_features(torch.randn(1, 16, 512, 512, generator=g))
calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
for R, octree in ((64, True), (128, True), (128, False), (256, True)):
    try:
        t = time.perf_counter()
        sdf, n = mesh_util.eval_grid_device(netMR, calib, R, 5000, octree)
        torch.cuda.synchronize()
        print(R, octree, 'evaluated', n, 'sdf min/max', float(sdf.min()), float(sdf.max()), 'frac>0.5', float((sdf > 0.5).double().mean()), 'time', time.perf_counter() - t, flush=True)
        v, f = mesh_util.marching_cubes_device(netMR._ctx, sdf, 0.5)
        torch.cuda.synchronize()
        print('  MC', v.shape, f.shape, flush=True)
    except Exception as e:
        traceback.print_exc()
