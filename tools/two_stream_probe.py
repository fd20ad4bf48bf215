"""Would overlapping the batches of two streams pay?  Two contexts (two net pairs) evaluate the two halves of the 256^3
lattice concurrently from two host threads on two streams, against one context evaluating all of it."""
import sys, time, threading, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import _lib, nets
from pifuhd_amd.nets import _ptr
from helpers import mlp_only_nets, calib_tensor
sd, oc, of = mlp_only_nets()
def build():
    netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda'); netMR = nets.PIFuMRNet(of, netG); netG.eval()
    netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}; netG._push_weights()
    netMR._sd = {k: t.cuda() for k, t in sd.items() if not k.startswith('netG.')}; netMR._push_weights()
    g = torch.Generator().manual_seed(4242)
    netG.set_features(torch.randn(1, 256, 128, 128, generator=g)); netMR.set_features(torch.randn(1, 16, 512, 512, generator=g))
    _lib.call('pifu_reserve', netMR._ctx.h, int(sys.argv[1]) if len(sys.argv) > 1 else 1048576)
    return netMR
A, B = build(), build()
calib = np.ascontiguousarray(calib_tensor()[0].numpy()); cinv = np.ascontiguousarray(np.linalg.inv(calib.astype(np.float64)))
R = 256; res = (C.c_int32 * 3)(R, R, R); total = R ** 3; chunk = 5000
half = (total // 2 // chunk) * chunk
out = torch.empty(total, device='cuda')
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def run(net, first, count, stream):
    _lib.call('pifu_eval_lattice', net._ctx.h, res, cinv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p), None,
              first, count, chunk, _lib.Q_FULL_COARSE, C.c_void_p(out.data_ptr() + 4 * first), C.c_void_p(stream.cuda_stream))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(A, 0, total, sA); torch.cuda.synchronize(); t1 = time.perf_counter()
    ta = threading.Thread(target=run, args=(A, 0, half, sA)); tb = threading.Thread(target=run, args=(B, half, total - half, sB))
    ta.start(); tb.start(); ta.join(); tb.join(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('one stream %.1f ms (%.1f Mq/s)   two streams %.1f ms (%.1f Mq/s)' % ((t1 - t0) * 1e3, total / (t1 - t0) / 1e6, (t2 - t1) * 1e3, total / (t2 - t1) / 1e6), flush=True)
