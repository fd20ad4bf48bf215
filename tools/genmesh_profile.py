#!/usr/bin/env python3
"""gen_mesh at one resolution, one subject, for the profiler and for DESIGN.md: per call the executed FLOPs behind bench.py's
gen_mesh.roofline and the phase timings, then a finer breakdown of reconstruction().

    python tools/genmesh_profile.py 512 [--subject headline|body|pt_body] [--batch-rows 2097152]

--batch-rows: the point rows per batch of launches (bench.py's default: 2 097 152).  The number of k_mlp_gemm_tri launches per
call follows from it (round 5 quoted 244 launches from this tool at the library's default of 1 M rows beside bench.py's 144 at
2 M): it is printed with every launch count."""
import argparse
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, '.')
import torch                                     # noqa: E402
import pifuhd_amd                                # noqa: E402,F401
from pifuhd_amd import _lib, config, mesh_util, nets, reconstruction as recon, synth      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('res', type=int, nargs='?', default=512)
ap.add_argument('--subject', default='headline', choices=['headline', 'body', 'pt_body'])
ap.add_argument('--batch-rows', type=int, default=2097152)
ap.add_argument('--calls', type=int, default=2)
ap.add_argument('--plain', type=int, default=0, help='also time N calls WITHOUT timings / profiling (no phase-boundary synchronisations): the cost of the instrumentation')
args = ap.parse_args()
RES = args.res
pt = args.subject == 'pt_body'
oc = config.coarse_opt(in_image_channels=6, mlp_norm='no' if pt else 'group', norm='group' if pt else 'batch')
of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.10 if pt else 0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0, kind='texture' if args.subject == 'headline' else 'body')
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
if args.subject == 'headline':
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
else:
    sd, fit = synth.craft_body_mlps(sd, oc, of, netG.im_feat_list[-1])
    netMR.load_state_dict(sd)
    print('crafted body subject:', fit)
_lib.call('pifu_reserve', netMR._ctx.h, args.batch_rows)
print(f'subject {args.subject}, {RES}^3, batch_rows {args.batch_rows}')


def prof(which, reset):
    ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
    _lib.call('pifu_profile_read', netMR._ctx.h, which, C.byref(ms), C.byref(n), C.byref(w), reset)
    return ms.value, n.value, w.value


with tempfile.TemporaryDirectory() as td:
    recon.gen_mesh(128, netMR, 'cuda', data, os.path.join(td, 'w.obj'))
    recon.gen_mesh(RES, netMR, 'cuda', data, os.path.join(td, 'w2.obj'))          # first call at the size: allocations
    for k in range(args.calls):
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
        print('call %d: k_mlp_gemm_tri %d launches (batch_rows %d), %.3f TFLOP, %.1f ms summed (HIP events); encoder convolutions %d launches, '
              '%.3f TFLOP; (%.3f + %.3f) TFLOP / %.4f s / 157.3 TFLOP/s = %.4f of the fp32-matrix peak' %
              (k, g_n, args.batch_rows, g_fl / 1e12, g_ms, e_n, e_fl / 1e12, g_fl / 1e12, e_fl / 1e12, t['total_s'],
               (g_fl + e_fl) / t['total_s'] / 157.3e12), flush=True)
    if args.plain:
        for mode in ('instrumented', 'plain', 'instrumented', 'plain'):
            ts = []
            for k in range(args.plain):
                path = os.path.join(td, 'p%d.obj' % k)
                if mode == 'instrumented':
                    _lib.call('pifu_profile_enable', netMR._ctx.h, 1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                recon.gen_mesh(RES, netMR, 'cuda', data, path, timings={} if mode == 'instrumented' else None)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                _lib.call('pifu_profile_enable', netMR._ctx.h, 0)
                os.remove(path)
            ts.sort()
            print('%-12s gen_mesh wall clock: median %.4f s, min %.4f s (%d calls)' % (mode, ts[len(ts) // 2], ts[0], len(ts)), flush=True)
    # finer breakdown of reconstruction()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sdf, n = mesh_util.eval_grid_device(netMR, data['calib'], RES, 5000, True); torch.cuda.synchronize(); t1 = time.perf_counter()
    v, f = mesh_util.marching_cubes_device(netMR._ctx, sdf, 0.5); torch.cuda.synchronize(); t2 = time.perf_counter()
    vh = v.cpu().numpy(); fh = f.cpu().numpy(); t3 = time.perf_counter()
    print('octree eval %.3f s (%d points evaluated = %.2f%% of grid), MC %.3f s, D2H %.3f s' % (t1 - t0, n, 100.0 * n / RES ** 3, t2 - t1, t3 - t2))
