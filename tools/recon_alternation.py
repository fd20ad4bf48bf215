"""Why does reconstruction_s of consecutive gen_mesh calls alternate between ~0.29 and ~0.33 s?  Per call: the wall time of the
octree evaluation alone, the GEMM kernel time inside it (HIP events of the library) and the shader clock over it."""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd
from pifuhd_amd import _lib, config, nets, synth, mesh_util, reconstruction as recon
from pifuhd_amd.nets import _stream_ptr
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
h = netMR._ctx.h
dev = netMR.device
def probe():
    buf = (C.c_uint64 * 16)(); _lib.call('pifu_clock_probe', h, buf, _stream_ptr(dev)); return list(buf)
def mhz(a, b):
    m = sorted(100.0 * (b[2*x] - a[2*x]) / (b[2*x+1] - a[2*x+1]) for x in range(8) if a[2*x+1] and b[2*x+1] > a[2*x+1])
    return m[len(m)//2] if m else 0
def gemm_ms():
    ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
    _lib.call('pifu_profile_read', h, _lib.PROF_GEMM, C.byref(ms), C.byref(n), C.byref(w), 1); return ms.value, n.value
with tempfile.TemporaryDirectory() as td:
    recon.gen_mesh(512, netMR, 'cuda', data, os.path.join(td, 'w.obj'))
    for k in range(8):
        t = {}
        recon.gen_mesh(512, netMR, 'cuda', data, os.path.join(td, 'm%d.obj' % k), timings=t)
        os.remove(os.path.join(td, 'm%d.obj' % k))
        # the evaluation alone, right after that call, instrumented
        _lib.call('pifu_profile_enable', h, 1); gemm_ms()
        torch.cuda.synchronize(); c0 = probe(); t0 = time.perf_counter()
        sdf, n = mesh_util.eval_grid_device(netMR, data['calib'], 512, 5000, True)
        torch.cuda.synchronize(); t1 = time.perf_counter(); c1 = probe()
        g, gn = gemm_ms(); _lib.call('pifu_profile_enable', h, 0)
        print('call %d: gen_mesh reconstruction_s %.4f normals_s %.4f | eval alone %.4f s, GEMM %.1f ms in %d launches, clock %.0f MHz, sdf at %x' % (
            k, t['reconstruction_s'], t['normals_s'], t1 - t0, g, gn, mhz(c0, c1), sdf.data_ptr()), flush=True)
        del sdf
