"""Sweep of the pixel-table path against the per-point GEMM path (pifu_set_pixel_tables 2 vs 0) over MLP norm modes, feature-map
sizes (square, non-square, pixel counts that are not a multiple of the 128-row tile), calibrations (zoom + shift, rotations
about x / y / z) and point clouds that reach outside the box: largest occupancy difference, mask equality, coarse predictions,
and calc_normal's occupancies.  Prints one line per case and fails on a difference above 2e-5 x the norm mode's scale."""
import ctypes as C
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
import pifuhd_amd  # noqa: F401
from pifuhd_amd import _lib, nets
from pifuhd_amd.nets import _ptr, _stream_ptr
from helpers import golden_nets

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
worst = 0.0


def calib_of(kind, rng):
    c = np.eye(4, dtype=np.float32)
    zoom = rng.uniform(0.8, 1.3)
    if kind == 'plain':
        c[1, 1] = -1.0
    elif kind == 'zoom':
        c[0, 0], c[1, 1], c[2, 2] = zoom, -zoom, zoom
        c[:3, 3] = rng.uniform(-0.1, 0.1, 3)
    else:
        a = math.radians(rng.uniform(5, 40))
        i, j = {'rot_x': (1, 2), 'rot_y': (0, 2), 'rot_z': (0, 1)}[kind]
        r = np.eye(3, dtype=np.float32)
        r[i, i], r[i, j], r[j, i], r[j, j] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
        c[:3, :3] = zoom * r @ np.diag([1.0, -1.0, 1.0]).astype(np.float32)
        c[:3, 3] = rng.uniform(-0.05, 0.05, 3)
    return c


for mlp_norm, scale in (('group', 1.0), ('batch', 3.0), ('no', 3.0)):
    sd, oc, of = golden_nets(mlp_norm=mlp_norm)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    h = netMR._ctx.h
    for case, (hc, wc, hf, wf) in enumerate(((32, 32, 64, 64), (20, 20, 40, 40), (24, 40, 48, 80), (128, 128, 512, 512))):
        rng = np.random.default_rng(100 * case + len(mlp_norm))
        g = torch.Generator().manual_seed(7 + case)
        netG.set_features(torch.randn(1, 256, hc, wc, generator=g))
        netMR.set_features(torch.randn(1, 16, hf, wf, generator=g))
        for kind in ('plain', 'zoom', 'rot_x', 'rot_y', 'rot_z'):
            cal = np.ascontiguousarray(calib_of(kind, rng))
            pts = ((torch.rand(3, N, generator=g) * 2 - 1) * 1.1).cuda().contiguous()
            res = {}
            for mode in (_lib.PIX_OFF, _lib.PIX_ALWAYS):
                _lib.call('pifu_set_pixel_tables', h, mode)
                fine, coarse = torch.empty(N, device='cuda'), torch.empty(N, device='cuda')
                _lib.call('pifu_query', h, _ptr(pts), N, cal.ctypes.data_as(C.c_void_p), 5000, 0, _ptr(fine), _ptr(coarse), None,
                          _stream_ptr(netMR.device))
                nv = N // 8
                nm, p4 = torch.empty(3, nv, device='cuda'), torch.empty(4 * nv, device='cuda')
                _lib.call('pifu_calc_normal', h, _ptr(pts[:, :nv].contiguous() * 0.8), nv, cal.ctypes.data_as(C.c_void_p), 0.001, _ptr(nm),
                          _ptr(p4), _stream_ptr(netMR.device))
                res[mode] = (fine.cpu().numpy(), coarse.cpu().numpy(), p4.cpu().numpy())
            a, b = res[_lib.PIX_OFF], res[_lib.PIX_ALWAYS]
            d = [float(np.abs(x - y).max()) for x, y in zip(a, b)]
            same_mask = bool(np.array_equal(a[0] == 0, b[0] == 0) and np.array_equal(a[1] == 0, b[1] == 0))
            ran = not np.array_equal(a[0], b[0])
            print(f'{mlp_norm:5s} map {hc:3d}x{wc:3d} {kind:6s}: fine {d[0]:.2e} coarse {d[1]:.2e} calc_normal occupancies {d[2]:.2e} '
                  f'masks equal {same_mask} other path ran {ran}', flush=True)
            assert same_mask and ran and max(d) < 2e-5 * scale, (mlp_norm, case, kind, d)
            worst = max(worst, max(d) / scale)
    _lib.call('pifu_set_pixel_tables', h, _lib.PIX_AUTO)
print('largest difference (per norm-mode scale):', worst)
