"""Where the host-side time of gen_mesh 512^3 goes (copies, numpy transforms) between the GPU phases."""
import sys, time, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pifuhd_amd
from pifuhd_amd import config, nets, synth, reconstruction as recon, mesh_util
from numpy.linalg import inv
oc = config.coarse_opt(in_image_channels=6); of = config.fine_opt(oc)
sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
data = synth.synthetic_inputs(1024, 6, seed=0)
netMR.filter_global(data['img_512'].cuda()); netMR.filter_local(data['img'].cuda()[:, None])
synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
calib = data['calib'].cuda()
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(2):
    t = [T()]
    sdf, n = mesh_util.eval_grid_device(netMR, calib, 512, 5000, True); t.append(T())
    v_d, f_d = mesh_util.marching_cubes_device(netMR._ctx, sdf, 0.5); t.append(T())
    verts = v_d.cpu().numpy(); t.append(T())
    faces = f_d.cpu().numpy(); t.append(T())
    ci = inv(calib[0].cpu().numpy().astype(np.float64)); mat = np.eye(4); mat[0, 0] = mat[1, 1] = mat[2, 2] = 2.0 / 512; mat[0:3, 3] = -1
    tm = ci @ mat
    verts64 = (np.matmul(tm[:3, :3], verts.T) + tm[:3, 3:4]).T; t.append(T())
    vt = torch.from_numpy(verts64.T).unsqueeze(0).to(device='cuda').float(); t.append(T())
    netMR.calc_normal_chunked(vt[:, None, :, :len(verts64) - 1], calib[:, None], calib, interval=50000); t.append(T())
    col = (netMR.nmls.detach()[0] * 0.5 + 0.5).T.cpu().numpy(); t.append(T())
    color = np.zeros(verts64.shape); color[:col.shape[0]] = col; t.append(T())
    with tempfile.TemporaryDirectory() as td:
        mesh_util.save_obj_mesh_with_color(os.path.join(td, 'm.obj'), verts64, faces[:, ::-1], color); t.append(T())
    names = ['octree eval', 'MC', 'verts D2H', 'faces D2H', 'vertex transform (numpy f64)', 'verts H2D + cast', 'calc_normal (GPU)', 'colours D2H', 'colour array', 'OBJ write']
    print({n: round(b - a, 4) for n, a, b in zip(names, t[:-1], t[1:])}, 'total', round(t[-1] - t[0], 3))
