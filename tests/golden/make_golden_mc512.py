#!/opt/conda/bin/python3.9
"""Marching-cubes golden for BASELINE config 3's grid, with scikit-image 0.18.3 (the pinned third-party
`skimage.measure.marching_cubes_lewiner` of `mesh_util.py:84`):

    /opt/conda/bin/python3.9 tests/golden/make_golden_mc512.py

Consumes /tmp/golden_sdf_oct512.npy (the float64 SDF the REFERENCE's eval_grid_octree produced at R = 512 on the analytic
noisy field, written by make_golden_r2.py octree512) and writes tests/golden/mc_oct512.npz: vertex / face counts, sha256 of
both streams as skimage returns them, and strided samples for a readable failure.
"""
import hashlib
import os
import warnings

import numpy as np

warnings.simplefilter('ignore')
import skimage  # noqa: E402
from skimage import measure  # noqa: E402

assert skimage.__version__ == '0.18.3', skimage.__version__
HERE = os.path.dirname(os.path.abspath(__file__))
sdf = np.load('/tmp/golden_sdf_oct512.npy')
v, f, _, _ = measure.marching_cubes_lewiner(sdf, 0.5)
v = np.ascontiguousarray(v, dtype=np.float32)
f = np.ascontiguousarray(f, dtype=np.int32)
out = {'nv_nf': np.array([len(v), len(f)]),
       'v_sha': np.frombuffer(hashlib.sha256(v.tobytes()).digest(), np.uint8),
       'f_sha': np.frombuffer(hashlib.sha256(f.tobytes()).digest(), np.uint8),
       'v_sub': v[::997].copy(), 'f_sub': f[::997].copy()}
np.savez_compressed(os.path.join(HERE, 'mc_oct512.npz'), **out)
print('mc_oct512.npz: V, F =', len(v), len(f))
