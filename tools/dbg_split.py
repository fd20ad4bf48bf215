import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import _lib, nets
from helpers import golden_nets, seeded_features, seeded_points, calib_tensor
sd, oc, of = golden_nets()
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
fc, ff = seeded_features(7); netG.set_features(fc); netMR.set_features(ff)
calib = calib_tensor()
pts = seeded_points(3, 700)
netMR.query(pts, calib); exact = netMR.get_preds()[0,0].cpu().numpy()
for mode in (1, 2, 1):
    _lib.call('pifu_set_precision', netMR._ctx.h, mode)
    netMR.query(pts, calib); s = netMR.get_preds()[0,0].cpu().numpy()
    print('mode', mode, 'max diff', np.abs(s-exact).max(), 'nan', np.isnan(s).sum(), 'coarse diff', None)
    _lib.call('pifu_set_precision', netMR._ctx.h, 0)
