#!/opt/conda/bin/python3.9
"""Generate the marching-cubes golden fixtures with scikit-image 0.18.3 (the version this build pins for
the reference's un-vendored `skimage.measure.marching_cubes_lewiner`, `mesh_util.py:84`).

    /opt/conda/bin/python3.9 tests/golden/make_golden_mc.py

Runs only in the build container.  Writes tests/golden/mc_small.npz (volumes + verts + faces).
Optionally consumes /tmp/golden_sdf_*.npy written by make_golden.py (reference-network SDFs).
"""
import hashlib
import os
import sys
import warnings

import numpy as np

warnings.simplefilter('ignore')
from skimage import measure  # noqa: E402
import skimage  # noqa: E402

assert skimage.__version__ == '0.18.3', skimage.__version__
HERE = os.path.dirname(os.path.abspath(__file__))


def mc(vol, level=0.5):
    v, f, _, _ = measure.marching_cubes_lewiner(vol, level)
    return np.ascontiguousarray(v, dtype=np.float32), np.ascontiguousarray(f, dtype=np.int32)


def smooth(v):
    out = v.copy()
    for ax in range(3):
        out = (np.roll(out, 1, ax) + out + np.roll(out, -1, ax)) / 3.0
    return out


out = {}
# ---- 1. known-answer vectors
kat = []
v = np.zeros((2, 2, 2), np.float32); v[0, 0, 0] = 1; kat.append(v)
v = np.zeros((2, 2, 2), np.float32); v[1, 0, 0] = 0.9; kat.append(v)
v = np.zeros((2, 2, 2), np.float32); v[0, 0, 0] = 0.5; v[1, 1, 1] = 1; kat.append(v)
v = np.zeros((4, 2, 2), np.float32); v[0, 0, 0] = 1; v[3, 0, 0] = 1; kat.append(v)
v = np.zeros((2, 4, 2), np.float32); v[0, 0, 0] = 1; v[0, 3, 0] = 1; kat.append(v)
v = np.zeros((2, 2, 4), np.float32); v[0, 0, 0] = 1; v[0, 0, 3] = 1; kat.append(v)
v = np.zeros((2, 2, 2), np.float32); v[0, 0, 0] = 1; v[0, 1, 1] = 1; kat.append(v)
v = np.zeros((2, 2, 2), np.float32); v[0, 0, 0] = 0.51; v[0, 1, 1] = 0.51; kat.append(v)
out['kat_n'] = np.array([len(kat)])
for i, vol in enumerate(kat):
    vv, ff = mc(vol)
    out[f'kat{i}_vol'], out[f'kat{i}_v'], out[f'kat{i}_f'] = vol, vv, ff

# ---- 2. random noise volumes (cover the ambiguous cases 3,4,6,7,10,12,13)
rs = np.random.RandomState(1234)
shapes = [(5, 5, 5), (6, 7, 5), (7, 7, 7), (8, 6, 9), (9, 9, 9), (11, 11, 11), (10, 4, 12), (3, 13, 6)]
n = 0
for shp in shapes:
    for kind in ('raw', 'smooth'):
        vol = rs.rand(*shp)
        if kind == 'smooth':
            vol = smooth(vol)
            vol = (vol - vol.min()) / (vol.max() - vol.min())
        vol = vol.astype(np.float32)
        vv, ff = mc(vol)
        out[f'noise{n}_vol'], out[f'noise{n}_v'], out[f'noise{n}_f'] = vol, vv, ff
        n += 1
out['noise_n'] = np.array([n])

# ---- 3. single cells, tie-heavy one-decimal values (exercise the eps / >= branches)
cells = (rs.randint(0, 11, size=(4000, 2, 2, 2)) / 10.0).astype(np.float32)
cv, cf, cnv, cnf = [], [], [], []
for c in cells:
    try:
        vv, ff = mc(c)
    except (ValueError, RuntimeError):
        vv, ff = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
    cv.append(vv); cf.append(ff); cnv.append(len(vv)); cnf.append(len(ff))
out['cells_vol'] = cells
out['cells_v'] = np.concatenate(cv)
out['cells_f'] = np.concatenate(cf)
out['cells_nv'] = np.array(cnv, np.int32)
out['cells_nf'] = np.array(cnf, np.int32)

# ---- 4. a 40^3 noisy ellipsoid (many cells, mixed cases); volume stored
lin = -1 + 2 * np.arange(40) / 40.0
X, Y, Z = np.meshgrid(lin, lin, lin, indexing='ij')
r = np.sqrt((0.8 * X) ** 2 + (0.45 * Y) ** 2 + (1.6 * Z) ** 2)
vol = 1.0 / (1.0 + np.exp(-(0.35 - r) / 0.05)) + 0.15 * (rs.rand(40, 40, 40) - 0.5)
vol = vol.astype(np.float32)
vv, ff = mc(vol)
out['ell40_vol'], out['ell40_v'], out['ell40_f'] = vol, vv, ff

# ---- 5. reference-network SDFs captured by make_golden.py (volumes live in recon_small.npz)
for tag in ('dense32', 'oct64'):
    p = f'/tmp/golden_sdf_{tag}.npy'
    if os.path.exists(p):
        sdf = np.load(p)
        vv, ff = mc(sdf)
        out[f'{tag}_nv_nf'] = np.array([len(vv), len(ff)])
        out[f'{tag}_v_sha'] = np.frombuffer(hashlib.sha256(vv.tobytes()).digest(), np.uint8)
        out[f'{tag}_f_sha'] = np.frombuffer(hashlib.sha256(ff.tobytes()).digest(), np.uint8)
        if tag == 'dense32':
            out[f'{tag}_v'], out[f'{tag}_f'] = vv, ff
    else:
        print('skip', p, file=sys.stderr)

np.savez_compressed(os.path.join(HERE, 'mc_small.npz'), **out)
print('mc_small.npz written:', len(out), 'arrays; cells with surface:', int((np.array(cnv) > 0).sum()),
      'ell40 V,F =', len(out['ell40_v']), len(out['ell40_f']))
