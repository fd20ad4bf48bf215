"""reconstruction_s of consecutive gen_mesh calls with and without the debug-PNG thread running beside it."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, reconstruction as recon
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
orig = recon._save_debug_png
with tempfile.TemporaryDirectory() as td:
    recon.gen_mesh(512, netMR, 'cuda', data, os.path.join(td, 'w.obj'))
    for mode in ('png thread', 'no png', 'png thread', 'no png'):
        recon._save_debug_png = orig if mode == 'png thread' else (lambda *a, **k: None)
        for k in range(4):
            t = {}
            recon.gen_mesh(512, netMR, 'cuda', data, os.path.join(td, 'm.obj'), timings=t)
            os.remove(os.path.join(td, 'm.obj'))
            print(mode, {k_: round(v, 4) for k_, v in t.items() if k_.endswith('_s')}, flush=True)
