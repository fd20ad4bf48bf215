"""The per-column tables of the lattice evaluations stay in the context between calls (ABI 5: the levels of one octree
evaluation share them): every change of an input must drop them, and keeping them must not change a bit."""
import numpy as np
import pytest
import torch

from helpers import calib_tensor, golden_nets, seeded_features

pytestmark = pytest.mark.gpu


def _net(feat_seed):
    from pifuhd_amd import nets
    sd, oc, of = golden_nets()
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(feat_seed)
    netG.set_features(fc)
    netMR.set_features(ff)
    return netG, netMR


def test_column_tables_follow_their_inputs():
    from pifuhd_amd import _lib, mesh_util, synth
    calib = calib_tensor()
    netG, netMR = _net(7)
    ev = lambda net: mesh_util.eval_grid_device(net, calib, 64, 5000, True, init_resolution=16)[0].clone()     # noqa: E731
    a1 = ev(netMR)                              # three octree levels: the first builds the tables, the others reuse them
    a2 = ev(netMR)                              # a second call on the same inputs reuses them too
    _lib.call('pifu_drop_tables', netMR._ctx.h)
    a3 = ev(netMR)                              # rebuilt from scratch
    assert torch.equal(a1, a2) and torch.equal(a1, a3)
    # other feature maps: the tables of the previous maps must not be used
    fc2, ff2 = seeded_features(8)
    netG.set_features(fc2)
    b_coarse_only = ev(netMR)
    netMR.set_features(ff2)
    b = ev(netMR)
    _, fresh = _net(8)
    assert torch.equal(b, ev(fresh)) and not torch.equal(b, b_coarse_only) and not torch.equal(a1, b_coarse_only)
    # other weights (the bias calibration reloads the fine MLP), another lattice, the precision switch
    shift = synth.calibrate_occupancy_bias(netMR, calib, dict(netMR._sd), 0.3, resolution=32)
    c = ev(netMR)
    assert abs(shift) > 1e-3 and not torch.equal(c, b)
    d48 = mesh_util.eval_grid_device(netMR, calib, 48, 5000, True, init_resolution=16)[0]
    assert d48.shape == (48, 48, 48) and torch.equal(ev(netMR), c)
    _lib.call('pifu_set_precision', netMR._ctx.h, 2)
    e = ev(netMR)
    _lib.call('pifu_set_precision', netMR._ctx.h, 0)
    assert not torch.equal(e, c) and float((e - c).abs().max()) < 1e-4 and torch.equal(ev(netMR), c)
