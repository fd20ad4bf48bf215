"""Marching-cubes time on a smooth 512^3 volume (the analytic ellipsoid of SURVEY 8d: ~250 k vertices) and on a noisy one."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd
from pifuhd_amd import _lib, mesh_util
if '--lib' in sys.argv:          # time another build of the library (tools/lab/r02: the round-2 kernels, for before / after)
    i = sys.argv.index('--lib')
    _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
from pifuhd_amd.nets import _Ctx
ctx = _Ctx('cuda')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.linspace(-1, 1, R, device='cuda', dtype=torch.float64)
x, y, z = torch.meshgrid(g, g, g, indexing='ij')
r = torch.sqrt((0.8 * x) ** 2 + (0.45 * y) ** 2 + (1.6 * z) ** 2)
smooth = torch.sigmoid((0.35 - r) / 0.01)
noisy = smooth + 0.2 * torch.sin(37 * x + 11 * y) * torch.cos(23 * z) * torch.rand_like(smooth)
for name, vol in (('smooth', smooth), ('noisy', noisy)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v, f = mesh_util.marching_cubes_device(ctx, vol, 0.5)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%s %d^3: %d verts, %d faces, %.2f ms' % (name, R, len(v), len(f), dt * 1e3))
