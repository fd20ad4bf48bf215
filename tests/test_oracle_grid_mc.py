"""Pin the oracle's grid evaluation (dense + octree) to the reference's `mesh_util` and its marching
cubes to scikit-image 0.18.3 (fixtures: tests/golden/octree_small.npz, mc_small.npz, recon_small.npz)."""
import hashlib
import os

import numpy as np
import pytest

from helpers import noisy_field
from oracle import pifu_oracle as O

CALIB_INV = np.linalg.inv(np.diag([1.0, -1.0, 1.0, 1.0]))


@pytest.fixture(scope='module')
def oct_gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'octree_small.npz'))


@pytest.fixture(scope='module')
def mc_gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'mc_small.npz'))


@pytest.mark.parametrize('R,init', [(32, 4), (64, 8), (64, 16), (64, 64), (48, 6)])
def test_octree_bit_exact(oct_gold, R, init):
    counts = []

    def f(p):
        counts.append(p.shape[1])
        return noisy_field(p)
    sdf = O.eval_grid_octree(R, CALIB_INV, f, 5000, init_resolution=init, threshold=0.05)
    assert hashlib.sha256(sdf.tobytes()).digest() == oct_gold[f'oct_{R}_{init}_sha'].tobytes()
    assert np.array_equal(sdf[::3, ::3, ::3], oct_gold[f'oct_{R}_{init}_sub'])
    assert np.array_equal(np.array(counts), oct_gold[f'oct_{R}_{init}_chunks'])   # same chunk membership


def test_dense_bit_exact(oct_gold):
    sdf = O.eval_grid(32, CALIB_INV, noisy_field, 5000)
    assert hashlib.sha256(sdf.tobytes()).digest() == oct_gold['dense_32_sha'].tobytes()


def _same(v, f, gv, gf):
    return (v.shape == gv.shape and f.shape == gf.shape and np.array_equal(f, gf)
            and np.array_equal(v.view(np.uint32), gv.view(np.uint32)))


def test_mc_known_answers_and_noise(mc_gold):
    g = mc_gold
    for i in range(int(g['kat_n'][0])):
        v, f = O.marching_cubes_lewiner(g[f'kat{i}_vol'], 0.5)
        assert _same(v, f, g[f'kat{i}_v'], g[f'kat{i}_f']), f'kat{i}'
    for i in range(int(g['noise_n'][0])):
        v, f = O.marching_cubes_lewiner(g[f'noise{i}_vol'], 0.5)
        assert _same(v, f, g[f'noise{i}_v'], g[f'noise{i}_f']), f'noise{i}'
    v, f = O.marching_cubes_lewiner(g['ell40_vol'], 0.5)
    assert _same(v, f, g['ell40_v'], g['ell40_f'])


def test_mc_tie_cells(mc_gold):
    g = mc_gold
    ov = np.concatenate([[0], np.cumsum(g['cells_nv'])])
    of = np.concatenate([[0], np.cumsum(g['cells_nf'])])
    for i, c in enumerate(g['cells_vol']):
        try:
            v, f = O.marching_cubes_lewiner(c, 0.5)
        except (ValueError, RuntimeError):
            v, f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
        assert _same(v, f, g['cells_v'][ov[i]:ov[i + 1]], g['cells_f'][of[i]:of[i + 1]]), f'cell {i}'


def test_mc_error_behaviour():
    with pytest.raises(ValueError):
        O.marching_cubes_lewiner(np.zeros((4, 4, 4), np.float32), 0.5)     # level outside data range
    vol = np.full((4, 4, 4), 0.5, np.float32)
    with pytest.raises(RuntimeError):
        O.marching_cubes_lewiner(vol, 0.5)                                   # in range but no surface


def test_mc_reference_network_sdfs(mc_gold, golden_dir):
    r = np.load(os.path.join(golden_dir, 'recon_small.npz'))
    for tag in ('dense32', 'oct64'):
        v, f = O.marching_cubes_lewiner(r[f'{tag}_sdf_f32'], 0.5)
        assert [len(v), len(f)] == list(mc_gold[f'{tag}_nv_nf'])
        assert hashlib.sha256(v.tobytes()).digest() == mc_gold[f'{tag}_v_sha'].tobytes()
        assert hashlib.sha256(f.tobytes()).digest() == mc_gold[f'{tag}_f_sha'].tobytes()
