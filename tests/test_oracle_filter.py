"""Pin the oracle's encoders (HGFilter, pix2pixHD normal nets) to the imported reference
(fixtures: tests/golden/filter_small.npz), including the train-mode BatchNorm quirk of the fine level
(SURVEY C-2) and the GroupNorm HGFilter variant."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_nets
from oracle import pifu_oracle as O
from pifuhd_amd import synth


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'filter_small.npz'))


def test_filter_global_and_local_tiny(gold):
    sd, oc, of = golden_nets()
    of.loadSizeBig = 128
    nets = O.OracleNets(sd, oc, of, fine_training=True)
    with torch.no_grad():
        fc = nets.filter_global(synth.synthetic_image(64, 3, seed=2))
        ff = nets.filter_local(synth.synthetic_image(128, 3, seed=3))
    assert np.abs(nets.nmlF.numpy() - gold['tiny_nmlF']).max() < 2e-5
    assert np.abs(nets.nmlB.numpy() - gold['tiny_nmlB']).max() < 2e-5
    scale = np.abs(gold['tiny_feat_coarse']).max()
    assert np.abs(fc.numpy() - gold['tiny_feat_coarse']).max() < 1e-4 * max(scale, 1.0)
    scale = np.abs(gold['tiny_feat_fine_trainmode']).max()
    assert np.abs(ff.numpy() - gold['tiny_feat_fine_trainmode']).max() < 1e-4 * max(scale, 1.0)


def test_fine_filter_eval_mode_differs(gold):
    """Running the fine HGFilter with running statistics instead of batch statistics is a different
    function (the reference never does it on the recon path)."""
    sd, oc, of = golden_nets()
    of.loadSizeBig = 128
    nets = O.OracleNets(sd, oc, of, fine_training=False)
    with torch.no_grad():
        nets.filter_global(synth.synthetic_image(64, 3, seed=2))
        ff = nets.filter_local(synth.synthetic_image(128, 3, seed=3))
    assert np.abs(ff.numpy() - gold['tiny_feat_fine_trainmode']).max() > 1e-2


def test_group_norm_filter(gold):
    spec = {}
    synth.filter_keys('image_filter', 1, 2, 9, 16, 'group', 'no_down', spec)
    sd = synth.name_seeded_state_dict(spec, seed=5, perturb=True)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 9, 64, 64, generator=g)
    with torch.no_grad():
        outs, _ = O.hg_filter(sd, 'image_filter', x, 1, 2, 'group', 'no_down', bn_training=False)
    assert np.abs(outs[-1].numpy() - gold['group_filter_out']).max() < 1e-4


def test_unknown_down_type_raises():
    with pytest.raises(NameError):
        synth.filter_keys('f', 1, 2, 9, 16, 'batch', 'conv64', {})
