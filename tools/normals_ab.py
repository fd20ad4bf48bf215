"""A/B of the calc_normal phase IN ONE PROCESS: the pixel tables of explicit point lists (pifu_set_pixel_tables, default) against
the per-point GEMM path, alternating.  Vertices: seeded points inside the box, in gen_mesh's 50 000-vertex calc_normal calls,
ten calls per launch group like the colour loop.  Prints ms per group, Mqueries/s, the executed GEMM TFLOP/s of the launches
(HIP events on the launch stream) and the largest difference between the two paths' occupancies."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd  # noqa: F401
from pifuhd_amd import _lib, config, nets, synth
from pifuhd_amd.nets import _ptr, _stream_ptr

NV = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
oc = config.coarse_opt(in_image_channels=6)
of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda())
netMR.filter_local(data['img'].cuda()[:, None])
h = netMR._ctx.h
_lib.call('pifu_reserve', h, 2097152)
g = torch.Generator().manual_seed(5)
verts = ((torch.rand(3, NV, generator=g) * 2 - 1) * 0.9).cuda().contiguous()
cal = data['calib'][0].numpy().astype('float32').copy()
dev = netMR.device


def run(mode, keep=False):
    _lib.call('pifu_set_pixel_tables', h, mode)
    out = torch.empty(3, NV, device='cuda')
    pred = torch.empty(4 * NV, device='cuda') if keep else None
    torch.cuda.synchronize()
    for w in (_lib.PROF_GEMM, _lib.PROF_SAMPLER):
        _lib.call('pifu_profile_read', h, w, None, None, None, 1)
    _lib.call('pifu_profile_enable', h, 1)
    t0 = time.perf_counter()
    _lib.call('pifu_calc_normal_chunked', h, _ptr(verts), NV, 50000, cal.ctypes.data_as(C.c_void_p), 0.001, _ptr(out), _ptr(pred),
              _stream_ptr(dev))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.call('pifu_profile_enable', h, 0)
    ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
    _lib.call('pifu_profile_read', h, _lib.PROF_GEMM, C.byref(ms), C.byref(n), C.byref(fl), 1)
    return dt, ms.value, n.value, fl.value, pred


run(_lib.PIX_AUTO)
run(_lib.PIX_OFF)
for rep in range(3):
    for mode, name in ((_lib.PIX_AUTO, 'pixel tables  '), (_lib.PIX_OFF, 'per-point GEMM')):
        dt, ms, n, fl, _ = run(mode)
        print(f'{name}: {dt * 1e3:8.2f} ms for {4 * NV} queries = {4 * NV / dt / 1e6:6.2f} Mq/s; GEMM launches {n}, {ms:7.2f} ms, '
              f'{fl / 1e12:6.2f} TFLOP executed = {fl / max(ms, 1e-9) / 1e9:6.1f} TFLOP/s', flush=True)
_, _, _, _, pa = run(_lib.PIX_AUTO, keep=True)
_, _, _, _, pb = run(_lib.PIX_OFF, keep=True)
_lib.call('pifu_set_pixel_tables', h, _lib.PIX_AUTO)
print('max |occupancy(pixel tables) - occupancy(per-point)| over', 4 * NV, 'queries:', float((pa - pb).abs().max()))
