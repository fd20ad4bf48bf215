"""Pin the CPU oracle to the round-2 goldens generated from the imported reference (tests/golden/make_golden_r2.py):
calc_normal's 4n occupancies at the 200 000-row batch gen_mesh uses, two non-trivial calibration matrices (query,
calc_normal and the dense reconstruction grid), and the 6-channel RGB-D encoder variant (SURVEY C-7)."""
import os

import numpy as np
import pytest
import torch

from helpers import calib_tensor, calib_variant, golden_nets, seeded_features, seeded_points
from oracle import pifu_oracle as O
from pifuhd_amd import config, synth

TOL_PRED = 5e-6


def test_oracle_calc_normal_occupancies_at_the_gen_mesh_batch_size(golden_dir):
    gold = np.load(os.path.join(golden_dir, 'normal_batch.npz'))
    n = int(gold['n'][0])
    sd, oc, of = golden_nets()
    nets = O.OracleNets(sd, oc, of, fine_training=True)
    nets.feat_coarse, nets.feat_fine = seeded_features(31, hc=128, hf=512)
    pts = seeded_points(41, n, lim=0.95)
    stack = [pts.clone() for _ in range(4)]
    for a in range(3):
        stack[a + 1][:, a, :] += 0.001
    with torch.no_grad():
        pred4 = nets.query(torch.stack(stack, 3).reshape(1, 3, 4 * n), calib_tensor(), mask_fine=False)[0, 0].numpy()
    assert np.abs(pred4[::7] - gold['pred4_sub']).max() < TOL_PRED
    assert np.abs(pred4[:4096] - gold['pred4_head']).max() < TOL_PRED


@pytest.mark.parametrize('tag', ['scale_shift', 'rot_y'])
def test_oracle_nontrivial_calibrations(golden_dir, tag):
    gold = np.load(os.path.join(golden_dir, 'calib_variants.npz'))
    sd, oc, of = golden_nets()
    nets = O.OracleNets(sd, oc, of, fine_training=True)
    nets.feat_coarse, nets.feat_fine = seeded_features(9)
    calib = calib_variant(tag)
    with torch.no_grad():
        fine = nets.query(seeded_points(12, 900, lim=1.1), calib)[0, 0].numpy()
        assert np.abs(fine - gold[f'{tag}_fine']).max() < TOL_PRED
        assert np.abs(nets.preds_coarse[0, 0].numpy() - gold[f'{tag}_coarse']).max() < TOL_PRED
        assert np.array_equal(fine == 0, gold[f'{tag}_fine'] == 0) and (fine == 0).sum() > 10
        nm = nets.calc_normal(seeded_points(13, 200, lim=0.8), calib)[0].numpy()
        cos = (nm * gold[f'{tag}_nml']).sum(0)
        assert np.median(cos) > 0.9999 and (cos > 0.99).mean() > 0.9
        # the dense grid of reconstruction(): float32 inverse of the calib, float64 grid, float32 points (mesh_util.py:59-70)
        cinv = np.linalg.inv(calib[0].numpy()).astype(np.float64)

        def field(p):
            return nets.query(torch.from_numpy(p.astype(np.float32))[None], calib)[0, 0].numpy()
        sdf = O.eval_grid(32, cinv, field, 5000)
    assert np.abs(sdf - gold[f'{tag}_dense32_sdf_f32']).max() < TOL_PRED


def test_oracle_rgbd_six_channel_encoders(golden_dir):
    gold = np.load(os.path.join(golden_dir, 'rgbd_small.npz'))
    oc = config.coarse_opt(in_image_channels=6)
    of = config.fine_opt(oc)
    of.loadSizeBig = 128
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=1, mlp_std=0.16, perturb=True)
    nets = O.OracleNets(sd, oc, of, fine_training=True)
    with torch.no_grad():
        fc = nets.filter_global(synth.synthetic_image(64, 6, seed=4))
        ff = nets.filter_local(synth.synthetic_image(128, 6, seed=5))
        assert np.abs(nets.nmlF.numpy() - gold['nmlF']).max() < 2e-5
        assert np.abs(nets.nmlB.numpy() - gold['nmlB']).max() < 2e-5
        assert np.abs(fc.numpy() - gold['feat_coarse']).max() < 1e-4 * max(np.abs(gold['feat_coarse']).max(), 1.0)
        assert np.abs(ff.numpy() - gold['feat_fine_trainmode']).max() < 1e-4 * max(np.abs(gold['feat_fine_trainmode']).max(), 1.0)
        fine = nets.query(seeded_points(14, 600), calib_tensor())[0, 0].numpy()
    assert np.abs(fine - gold['fine']).max() < TOL_PRED


def test_lattice_points_by_index_equal_the_grid():
    """mesh_util.lattice_points (the index-based point generator of the host-callback octree) against create_grid + the
    calib^-1 product of the reference, bit for bit (host-only code)."""
    from pifuhd_amd import mesh_util
    R = 12
    cinv = np.linalg.inv(calib_variant('rot_y')[0].numpy()).astype(np.float64)
    coords, _ = mesh_util.create_grid(R, R, R)
    c = coords.reshape(3, -1).T
    want = np.matmul(np.concatenate([c, np.ones((c.shape[0], 1))], 1), cinv.T)[:, :3].T
    idx = np.random.RandomState(0).permutation(R ** 3)
    assert np.array_equal(mesh_util.lattice_points(R, cinv, idx), want[:, idx])
    assert np.array_equal(O.grid_points(R, cinv, idx), want[:, idx])
