"""Wall clock of the encoder calls one by one and together (no profiler): filter_global alone (joined), filter_local alone,
the plain sequence, the overlapped schedule; images resident on the device."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import pifuhd_amd
from pifuhd_amd import config, nets, reconstruction, synth
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
img, img512 = data['img'].cuda(), data['img_512'].cuda()
dd = dict(data, img=img, img_512=img512)


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


netMR.filter_global(img512); netMR.filter_local(img[:, None])
print('filter_global alone          %.2f ms' % timed(lambda: netMR.filter_global(img512)))
print('filter_local alone           %.2f ms' % timed(lambda: netMR.filter_local(img[:, None])))
print('plain sequence               %.2f ms' % timed(lambda: (netMR.filter_global(img512), netMR.filter_local(img[:, None]))))
print('overlapped schedule          %.2f ms' % timed(lambda: reconstruction._Subject(netMR, dd)))
print('overlapped, images on host   %.2f ms' % timed(lambda: reconstruction._Subject(netMR, data)))
