"""A few per-point queries of 2 M random points (k_sample + the per-point MLP) for the sampler's PMC / trace passes."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import pifuhd_amd
from pifuhd_amd import nets
from helpers import mlp_only_nets, calib_tensor
sd, oc, of = mlp_only_nets()
netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda'); netMR = nets.PIFuMRNet(of, netG); netG.eval()
netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}; netG._push_weights()
netMR._sd = {k: t.cuda() for k, t in sd.items() if not k.startswith('netG.')}; netMR._push_weights()
g = torch.Generator().manual_seed(1)
netG.set_features(torch.randn(1, 256, 128, 128, generator=g)); netMR.set_features(torch.randn(1, 16, 512, 512, generator=g))
pts = (torch.rand(1, 3, 2_000_000, generator=g) * 2 - 1).cuda()
for _ in range(4):
    netMR.query(pts, calib_tensor())
torch.cuda.synchronize()
