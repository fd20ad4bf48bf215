"""Encoders alone (filter_global + filter_local of one 1024 px RGB-D subject): wall clock of the plain call sequence and of
the overlapped schedule gen_mesh uses (reconstruction._Subject), host image -> feature maps, H2D included.
Under rocprofv3 --kernel-trace --stats this is the source of profiles/rNN_encoder_kernel_stats.txt."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import pifuhd_amd
from pifuhd_amd import config, nets, reconstruction, synth
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for i in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
    torch.cuda.synchronize(); print('plain sequence, pass %d: %.2f ms' % (i, (time.perf_counter() - t0) * 1e3))
for i in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reconstruction._Subject(netMR, data)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); print('overlapped schedule (_Subject), pass %d: %.2f ms (host issue %.2f ms)' % (i, (time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3))
dimg, dimg512 = data['img'].cuda(), data['img_512'].cuda()
dd = dict(data, img=dimg, img_512=dimg512)
for i in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reconstruction._Subject(netMR, dd)
    torch.cuda.synchronize(); print('overlapped schedule, images already on the device, pass %d: %.2f ms' % (i, (time.perf_counter() - t0) * 1e3))
