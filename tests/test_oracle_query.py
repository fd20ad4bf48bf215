"""Pin the CPU oracle (oracle/pifu_oracle.py) to the golden vectors produced by importing the reference
(`tests/golden/make_golden.py`): query / get_preds / phi / calc_normal, three MLP norm modes."""
import os

import numpy as np
import pytest
import torch

from helpers import calib_tensor, golden_nets, seeded_features, seeded_points
from oracle import pifu_oracle as O

# fp32 CPU vs fp32 CPU with a different summation order (matmul vs conv1d): a few ulp after sigmoid
TOL_PRED = 5e-6
TOL_PHI = 1e-4          # phi is an un-squashed activation (|phi| up to ~10)


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'query_small.npz'))


@pytest.mark.parametrize('tag,mlp_norm', [('group', 'group'), ('batch', 'batch'), ('none', 'no')])
def test_oracle_query_matches_reference(gold, tag, mlp_norm):
    sd, oc, of = golden_nets(mlp_norm=mlp_norm)
    nets = O.OracleNets(sd, oc, of, fine_training=True)
    nets.feat_coarse, nets.feat_fine = seeded_features(7)
    calib = calib_tensor()
    scale = 1.0 if mlp_norm == 'group' else 20.0     # un-normalised variants amplify fp32 noise
    with torch.no_grad():
        fine = nets.query(seeded_points(3, 700), calib)
        assert np.abs(fine[0, 0].numpy() - gold[f'{tag}_a_fine']).max() < TOL_PRED * scale
        assert np.abs(nets.preds_coarse[0, 0].numpy() - gold[f'{tag}_a_coarse']).max() < TOL_PRED * scale
        assert np.abs(nets.phi[0, ::16].numpy() - gold[f'{tag}_a_phi_sub']).max() < TOL_PHI * scale
        # out-of-bbox points are exactly zero in both
        assert (gold[f'{tag}_a_fine'] == 0).sum() > 10
        assert np.array_equal(fine[0, 0].numpy() == 0, gold[f'{tag}_a_fine'] == 0)

        pts = seeded_points(4, 1200, lim=1.0)
        out = [nets.query(pts[:, :, s:s + 500], calib)[0, 0].numpy() for s in range(0, 1200, 500)]
        assert np.abs(np.concatenate(out) - gold[f'{tag}_b_fine']).max() < TOL_PRED * scale

        c, _ = nets.query_coarse(seeded_points(5, 300), calib)
        assert np.abs(c[0, 0].numpy() - gold[f'{tag}_c_coarse']).max() < TOL_PRED * scale


def test_oracle_chunk_coupling(gold):
    """GroupNorm statistics span the chunk (SURVEY C-1): evaluating the same points in one chunk of 1200
    instead of 500/500/200 must NOT reproduce the golden values."""
    sd, oc, of = golden_nets()
    nets = O.OracleNets(sd, oc, of)
    nets.feat_coarse, nets.feat_fine = seeded_features(7)
    with torch.no_grad():
        whole = nets.query(seeded_points(4, 1200, lim=1.0), calib_tensor())[0, 0].numpy()
    assert np.abs(whole - gold['group_b_fine']).max() > 1e-4


def test_oracle_calc_normal(gold):
    """Forward differences with delta=1e-3 amplify fp32 noise by ~1e3, so normals are compared by angle
    on the well-conditioned points only."""
    sd, oc, of = golden_nets()
    nets = O.OracleNets(sd, oc, of)
    nets.feat_coarse, nets.feat_fine = seeded_features(7)
    with torch.no_grad():
        n = nets.calc_normal(seeded_points(6, 150, lim=0.9), calib_tensor())[0].numpy()
    ref = gold['group_d_nml']
    cos = (n * ref).sum(0)
    assert np.median(cos) > 0.9999
    assert (cos > 0.99).mean() > 0.9
