"""Oracle end-to-end grid evaluation (dense 32^3 and octree 64^3, chunk 5000) with the network field,
against the SDF captured from the reference's `mesh_util.reconstruction` (tests/golden/recon_small.npz)."""
import os

import numpy as np
import torch

from helpers import calib_tensor, golden_nets, seeded_features
from oracle import pifu_oracle as O


def _eval_func(nets):
    calib = calib_tensor()

    def f(points):
        with torch.no_grad():
            p = torch.from_numpy(points[None]).float()
            return nets.query(p, calib)[0, 0].numpy()
    return f


def test_recon_sdf_matches_reference(golden_dir):
    gold = np.load(os.path.join(golden_dir, 'recon_small.npz'))
    sd, oc, of = golden_nets()
    nets = O.OracleNets(sd, oc, of)
    nets.feat_coarse, nets.feat_fine = seeded_features(9)
    calib_inv = np.linalg.inv(calib_tensor()[0].numpy().astype(np.float64))
    dense = O.eval_grid(32, calib_inv, _eval_func(nets), 5000)
    assert np.abs(dense - gold['dense32_sdf_f32']).max() < 5e-6
    octree = O.eval_grid_octree(64, calib_inv, _eval_func(nets), 5000)
    assert np.abs(octree - gold['oct64_sdf_f32']).max() < 5e-6
    # SURVEY C-3: the last index plane on every axis is never evaluated at R = 64
    assert np.all(octree[-1] == 0) and np.all(octree[:, -1] == 0) and np.all(octree[:, :, -1] == 0)
