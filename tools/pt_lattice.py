#!/usr/bin/env python3
"""The 256^3 lattice in the pifuhd.pt configuration (GroupNorm filters, MLP without norm layers), seeded random feature maps:
Mq/s and the GEMM launches' rate - a quick A/B for changes of the pointwise-MLP path (bench.py's pifuhd_pt_config.occupancy
is the number of record)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                     # noqa: E402
import pifuhd_amd                                # noqa: E402,F401
from pifuhd_amd import _lib, config, dist as pdist, nets, synth      # noqa: E402

oc = config.coarse_opt(in_image_channels=6, mlp_norm='no', norm='group')
of = config.fine_opt(oc)
spec = {}
synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
synth.mlp_keys('mlp', of.mlp_dim, of.mlp_res_layers, of.mlp_norm, spec)
full = synth.state_dict_spec(oc, of)
sd = {k: (synth.name_seeded_tensor(k, full[k], 0, mlp_std=0.10) if k in spec else torch.zeros(full[k])) for k in full}
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
g = torch.Generator().manual_seed(4242)
netG.set_features(torch.randn(1, 256, 128, 128, generator=g))
netMR.set_features(torch.randn(1, 16, 512, 512, generator=g))
calib = torch.eye(4)[None]
calib[0, 1, 1] = -1.0
h = netMR._ctx.h
_lib.call('pifu_reserve', h, 2097152)
res = (256, 256, 256)
flags = _lib.Q_FULL_COARSE


def step():
    _lib.call('pifu_drop_tables', h)
    return pdist.eval_lattice_sharded(netMR, calib, res, 5000, flags)


step(); torch.cuda.synchronize()
_lib.call('pifu_profile_enable', h, 1)
t0 = time.perf_counter()
for _ in range(3):
    f = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
_lib.call('pifu_profile_enable', h, 0)
ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
_lib.call('pifu_profile_read', h, _lib.PROF_GEMM, C.byref(ms), C.byref(n), C.byref(w), 1)
print('pifuhd.pt configuration, 256^3 lattice: %.1f Mq/s (%.2f ms per step); GEMM %d launches, %.1f TFLOP/s; field mean %.4f' %
      (256 ** 3 / dt / 1e6, dt * 1e3, n.value, w.value / (ms.value * 1e-3) / 1e12, float(f.mean())))
