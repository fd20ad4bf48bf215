"""A/B of gen_mesh IN ONE PROCESS (boxes of the pool differ by +-3 %): the vertex / face host copies deferred to the copy stream
(default) against waited for at once, alternating, distinct output paths."""
import sys, tempfile, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pifuhd_amd  # noqa: F401
from pifuhd_amd import config, nets, synth, reconstruction as recon
RES = int(sys.argv[1]) if len(sys.argv) > 1 else 512
std = float(sys.argv[2]) if len(sys.argv) > 2 else 0.16
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=std)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
with tempfile.TemporaryDirectory() as td:
    for k, mode in enumerate((True, False)):
        recon.gen_mesh(RES, netMR, 'cuda', data, os.path.join(td, 'w%d.obj' % k), _defer_host=mode)
    for rep in range(4):
        for mode in (True, False):
            t = {}
            p = os.path.join(td, 'm%d_%d.obj' % (rep, mode))
            recon.gen_mesh(RES, netMR, 'cuda', data, p, timings=t, _defer_host=mode)
            print('deferred copies' if mode else 'copies at once ', {k: round(v, 4) for k, v in t.items() if k.endswith('_s')}, flush=True)
            os.remove(p)
