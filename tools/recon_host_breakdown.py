"""Where the time of mesh_util.reconstruction() at 512^3 goes between its GPU phases (product functions, stepwise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, mesh_util
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
calib = data['calib'].cuda()
ctx = netMR._ctx
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(4):
    t = [T()]
    sdf, n = mesh_util.eval_grid_device(netMR, calib, 512, 5000, True); t.append(T())
    v_d, f_d = mesh_util.marching_cubes_device(ctx, sdf, 0.5); t.append(T())
    faces = mesh_util.to_host(f_d); t.append(T())
    _, cinv = mesh_util.calib_pair(calib)
    mat = np.eye(4); mat[0, 0] = mat[1, 1] = mat[2, 2] = 2.0 / 512; mat[0:3, 3] = -1.0
    verts, verts_dev = mesh_util.transform_verts(ctx, v_d, np.matmul(cinv, mat)); t.append(T())
    del sdf; t.append(T())
    names = ['octree eval', 'MC', 'faces to host', 'transform + verts to host', 'free sdf']
    print({k: round(b - a, 4) for k, a, b in zip(names, t[:-1], t[1:])}, 'sum', round(t[-1] - t[0], 4), flush=True)
    t0 = T(); r = mesh_util.reconstruction(netMR, 'cuda', calib, 512, None, None, 0.5, use_octree=True, num_samples=5000, return_device_verts=True); t1 = T()
    print('reconstruction() as a whole: %.4f' % (t1 - t0), flush=True)
    del r, faces, verts, verts_dev, v_d, f_d
