"""gen_mesh at the small sizes (128^3, 256^3: meshes of 0.2 - 0.9 M vertices, the size a trained net produces at 512^3): phase
times of three calls each.  At these sizes the GPU part takes 60 - 150 ms and the host-side tails (PNG, OBJ) show."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd  # noqa: F401
from pifuhd_amd import config, nets, synth, reconstruction as recon
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
with tempfile.TemporaryDirectory() as td:
    for R in (128, 256):
        recon.gen_mesh(R, netMR, 'cuda', data, os.path.join(td, 'w%d.obj' % R))
        for k in range(3):
            t = {}
            recon.gen_mesh(R, netMR, 'cuda', data, os.path.join(td, 'm%d_%d.obj' % (R, k)), timings=t)
            print(R, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()}, flush=True)
