"""Where the first call of each kind spends its time in a fresh process (weights already in host memory): context creation,
weight registration, first / second encoder pass, first / second / third gen_mesh at the given resolution with its phases.

    python tools/cold_start.py [resolution = 512]
"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RES = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t00 = time.perf_counter()
import torch
import pifuhd_amd  # noqa: F401
from pifuhd_amd import config, nets, synth, reconstruction as recon
t_imp = time.perf_counter()
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
t_cuda = time.perf_counter()
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
data = synth.synthetic_inputs(1024, 6, seed=0)
t_sd = time.perf_counter()
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
torch.cuda.synchronize(); t_build = time.perf_counter()
netMR._sync_weights()           # registers the weights with the context (convolution weights as device pointers)
torch.cuda.synchronize(); t_push = time.perf_counter()
ts = []
for k in range(3):
    a = time.perf_counter()
    netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
    torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
torch.cuda.synchronize()
print(f'import torch + package {t_imp - t00:.2f} s; HIP init {t_cuda - t_imp:.2f} s; synthetic weights + image (not the library) {t_sd - t_cuda:.2f} s')
print(f'build_nets (state dict -> device tensors) {t_build - t_sd:.3f} s; weight registration with the context {t_push - t_build:.3f} s')
print('encoder passes:', ' '.join(f'{t:.3f}' for t in ts), 's')
with tempfile.TemporaryDirectory() as td:
    for k in range(3):
        tm = {}
        a = time.perf_counter()
        recon.gen_mesh(RES, netMR, 'cuda', data, os.path.join(td, 'm%d.obj' % k), timings=tm)
        torch.cuda.synchronize()
        print(f'gen_mesh {RES}^3 call {k}: {time.perf_counter() - a:.3f} s  ' +
              ' '.join(f'{p}={tm[p]:.3f}' for p in ('filter_s', 'png_s', 'reconstruction_s', 'normals_s', 'obj_s')), flush=True)
