import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import pifuhd_amd
from pifuhd_amd import config, nets, synth
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
    torch.cuda.synchronize(); print('filter pass %d: %.1f ms' % (i, (time.perf_counter() - t0) * 1e3))
