"""Per-level breakdown of the 512^3 octree evaluation of the bench subject: points, column segments (runs of consecutive
lattice indices inside one (i, j) column - what decides whether the depth walk applies), evaluation time with and without
the walk.  usage: python tools/octree_levels.py [res]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import _lib, config, nets, synth
from pifuhd_amd.mesh_util import _ctx_of, calib_pair
from pifuhd_amd.nets import _ptr, _stream_ptr
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
ctx = _ctx_of(netMR); dev = ctx.device; st = _stream_ptr(dev)
_lib.call('pifu_reserve', ctx.h, 2097152)
calib, calib_inv = calib_pair(data['calib'])
res_a = (C.c_int32 * 3)(R, R, R)
for rep in range(2):
    sdf = torch.empty(R, R, R, device=dev, dtype=torch.float64); notproc = torch.empty(R, R, R, device=dev, dtype=torch.uint8)
    _lib.call('pifu_octree_begin', ctx.h, R, _ptr(sdf), _ptr(notproc), st)
    reso = R // 64
    while reso > 0:
        cnt = C.c_int64(0)
        _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), None, 0, C.byref(cnt), st)
        n = cnt.value
        idx = torch.empty(max(n, 1), device=dev, dtype=torch.int64)
        _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), _ptr(idx), n, C.byref(cnt), st)
        col = idx // R
        nseg = int((col[1:] != col[:-1]).sum().item()) + 1 + int(((col[1:] == col[:-1]) & (idx[1:] != idx[:-1] + reso)).sum().item())
        out = torch.empty(n, device=dev, dtype=torch.float32)
        t = {}
        for name, flags in (('walk allowed', 0), ('no walk', _lib.Q_NO_DEPTH_WALK)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _lib.call('pifu_eval_lattice', ctx.h, res_a, calib_inv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p),
                      _ptr(idx), 0, n, 5000, flags, _ptr(out), st)
            torch.cuda.synchronize(); t[name] = time.perf_counter() - t0
        if rep:
            print('step %d: %9d points, %8d column runs (%.1f points per run): %.1f ms (%.1f Mq/s); without the walk %.1f ms' % (
                reso, n, nseg, n / max(nseg, 1), t['walk allowed'] * 1e3, n / t['walk allowed'] / 1e6, t['no walk'] * 1e3))
        _lib.call('pifu_octree_commit', ctx.h, R, reso, 0.05, _ptr(idx), _ptr(out), n, _ptr(sdf), _ptr(notproc), st)
        if reso <= 1:
            break
        reso //= 2
