#!/usr/bin/env python3
"""Golden fixture for `esti_color` (the back-side recolouring of `gen_mesh_imgColor_plus`, `reconstruction.py:180-229`).

The reference module cannot be imported as a whole (it needs cv2, trimesh and a web API client at import time), but
`esti_color` itself is plain numpy / Python: this script loads ONLY that function definition from the reference file (ast),
executes it in an empty namespace with numpy, and records its outputs on seeded inputs.  Runs in the build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_color.py

Inputs are built so that the function's tie rules matter: coordinates snapped to a coarse lattice (marching-cubes vertices
lie on grid lines, so equal |dy| and equal x are the norm, not the exception), a thin silhouette band 0 <= z < 0.001, back
vertices with fewer than ten candidates on one side or none at all.
"""
import ast
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = '/root/reference/reconstruction.py'


def load_esti_color():
    tree = ast.parse(open(SRC, encoding='utf-8').read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'esti_color']
    assert len(fn) == 1
    mod = ast.Module(body=fn, type_ignores=[])
    ns = {'np': np}
    exec(compile(mod, SRC, 'exec'), ns)
    return ns['esti_color']


class _T:                      # stands in for the torch tensor argument: esti_color only calls .cpu().numpy()
    def __init__(self, a):
        self.a = a

    def cpu(self):
        return self

    def numpy(self):
        return self.a


def case(seed, n, lattice, band_frac, back_frac):
    rs = np.random.RandomState(seed)
    xy = rs.randint(-lattice, lattice + 1, size=(n, 2)).astype(np.float32) / np.float32(lattice)
    xy += (rs.rand(n, 2) < 0.3) * (rs.rand(n, 2).astype(np.float32) - 0.5) * np.float32(0.05)      # some off-lattice points
    u = rs.rand(n)
    z = np.where(u < back_frac, -rs.rand(n) * 0.5 - 1e-4, np.where(u < back_frac + band_frac, rs.rand(n) * 0.000999, 0.001 + rs.rand(n) * 0.5))
    xyz = np.concatenate([xy, z[:, None].astype(np.float32)], 1).astype(np.float32)
    color = rs.randint(0, 256, size=(n, 4)).astype(np.uint16)
    color[:, 3] = 255
    return xyz, color


def main():
    esti = load_esti_color()
    out = {}
    cases = [(1, 400, 6, 0.15, 0.4), (2, 900, 10, 0.05, 0.5), (3, 60, 3, 0.5, 0.3), (4, 300, 4, 0.02, 0.6)]
    out['n_cases'] = np.array([len(cases)])
    for i, c in enumerate(cases):
        xyz, color = case(*c)
        got = esti(color.copy(), _T(xyz.T[None].copy()))
        out[f'c{i}_xyz'] = xyz
        out[f'c{i}_color_in'] = color
        out[f'c{i}_color_out'] = np.asarray(got, dtype=np.uint16)
    # no silhouette vertex on either side of a back vertex: the reference raises
    xyz = np.array([[0.0, 0.0, -0.1], [0.5, 0.5, 0.5]], np.float32)
    try:
        esti(np.zeros((2, 4), np.uint16), _T(xyz.T[None]))
        out['empty_raises'] = np.array([0])
    except Exception:
        out['empty_raises'] = np.array([1])
    np.savez_compressed(os.path.join(HERE, 'esti_color.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
