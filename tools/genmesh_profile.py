import sys, time, tempfile, os
RES = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sys.path.insert(0, '.')
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, reconstruction as recon, mesh_util
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
import ctypes as C
from pifuhd_amd import _lib


def prof(which, reset):
    ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
    _lib.call('pifu_profile_read', netMR._ctx.h, which, C.byref(ms), C.byref(n), C.byref(w), reset)
    return ms.value, n.value, w.value


with tempfile.TemporaryDirectory() as td:
    recon.gen_mesh(128, netMR, 'cuda', data, os.path.join(td, 'w.obj'))
    recon.gen_mesh(RES, netMR, 'cuda', data, os.path.join(td, 'w2.obj'))          # first call at the size: allocations
    for k in range(2):
        t = {}
        prof(_lib.PROF_GEMM, 1); prof(_lib.PROF_ENC, 1)
        _lib.call('pifu_profile_enable', netMR._ctx.h, 1)
        # a path of its own per run: replacing an existing 470 MB file costs its page frees (~70 ms) inside the final rename
        recon.gen_mesh(RES, netMR, 'cuda', data, os.path.join(td, 'm%d.obj' % k), timings=t)
        _lib.call('pifu_profile_enable', netMR._ctx.h, 0)
        g_ms, g_n, g_fl = prof(_lib.PROF_GEMM, 1)
        _, e_n, e_fl = prof(_lib.PROF_ENC, 1)
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()}, flush=True)
        # what bench.py's gen_mesh.roofline is made of: executed algorithmic FLOPs of this one call / its wall time / 157.3e12
        print('call %d: k_mlp_gemm_tri %d launches, %.3f TFLOP, %.1f ms summed (HIP events); encoder convolutions %d launches, %.3f TFLOP; '
              '(%.3f + %.3f) TFLOP / %.4f s / 157.3 TFLOP/s = %.4f of the fp32-matrix peak' %
              (k, g_n, g_fl / 1e12, g_ms, e_n, e_fl / 1e12, g_fl / 1e12, e_fl / 1e12, t['total_s'],
               (g_fl + e_fl) / t['total_s'] / 157.3e12), flush=True)
    # finer breakdown of reconstruction()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sdf, n = mesh_util.eval_grid_device(netMR, data['calib'], RES, 5000, True); torch.cuda.synchronize(); t1 = time.perf_counter()
    v, f = mesh_util.marching_cubes_device(netMR._ctx, sdf, 0.5); torch.cuda.synchronize(); t2 = time.perf_counter()
    vh = v.cpu().numpy(); fh = f.cpu().numpy(); t3 = time.perf_counter()
    print('octree eval %.3f s (%d points evaluated = %.1f%% of grid), MC %.3f s, D2H %.3f s' % (t1 - t0, n, 100.0 * n / RES ** 3, t2 - t1, t3 - t2))
