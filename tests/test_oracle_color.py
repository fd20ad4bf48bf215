"""Pin the oracle's `esti_color` (back-side recolouring of gen_mesh_imgColor_plus) to the fixture produced by executing the
REFERENCE's own function (tests/golden/make_golden_color.py), and check the component oracle on hand-made meshes."""
import os

import numpy as np
import pytest

from oracle import pifu_oracle as O


def test_esti_color_matches_the_reference_function(golden_dir):
    gold = np.load(os.path.join(golden_dir, 'esti_color.npz'))
    for i in range(int(gold['n_cases'][0])):
        got = O.esti_color(gold[f'c{i}_color_in'], gold[f'c{i}_xyz'])
        assert np.array_equal(got, gold[f'c{i}_color_out']), i
        changed = (gold[f'c{i}_color_in'] != gold[f'c{i}_color_out']).any(1)
        assert changed.sum() > 5 and np.array_equal(changed & (gold[f'c{i}_xyz'][:, 2] >= 0), np.zeros_like(changed))
    assert int(gold['empty_raises'][0]) == 1
    with pytest.raises(Exception):
        O.esti_color(np.zeros((2, 4), np.uint16), np.array([[0.0, 0.0, -0.1], [0.5, 0.5, 0.5]], np.float32))


def tetra(offset, base):
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float) + offset
    f = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]]) + base
    return v, f


def test_face_components_oracle():
    v1, f1 = tetra(np.array([0.0, 0, 0]), 0)
    v2, f2 = tetra(np.array([5.0, 0, 0]) * 1.0, 4)
    f_open = np.array([[8, 9, 10]])                       # a lone triangle: three boundary edges
    faces = np.concatenate([f1, f2, f_open])
    label, open_face = O.face_components(faces)
    assert list(label) == [0, 0, 0, 0, 4, 4, 4, 4, 8]
    assert list(open_face) == [0] * 8 + [1]
