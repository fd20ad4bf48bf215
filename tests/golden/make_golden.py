#!/usr/bin/env python3
"""Generate the torch-half golden fixtures by IMPORTING THE REFERENCE (`/root/reference`).

Runs only in the build container (the reference never travels to the GPU box); the outputs are the
small `tests/golden/*.npz` files committed next to this script.  Inputs are all seed-derived
(`rgb-d-pifuhd_amd/synth.py`), so the fixtures hold expected OUTPUTS only (plus the seeds).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Stubs: `torchvision` (imported by `networks.py:418`) and `skimage.measure` (imported by
`mesh_util.py:10`) are absent from this interpreter and are replaced by empty modules; marching
cubes is handled by `make_golden_mc.py` under /opt/conda python3.9 (scikit-image 0.18.3).
"""
import copy
import hashlib
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
for name in ('torchvision', 'torchvision.models', 'torchvision.transforms', 'skimage', 'skimage.measure'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['torchvision'].models = sys.modules['torchvision.models']
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
sys.modules['skimage'].measure = sys.modules['skimage.measure']

import numpy as np   # noqa: E402
import torch         # noqa: E402

import pifuhd_amd    # noqa: E402  (shim -> rgb-d-pifuhd_amd/)
from pifuhd_amd import synth  # noqa: E402

from options import BaseOptions        # noqa: E402  (reference)
from PIFuNetwNML import PIFuNetwNML    # noqa: E402
from PIFuMRNet import PIFuMRNet        # noqa: E402
import mesh_util as ref_mesh_util      # noqa: E402
from Filter import Filter              # noqa: E402
from networks import define_G          # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def build_reference(mlp_norm='group', seed=1, mlp_std=0.16, perturb=True, norm='batch'):
    if mlp_norm not in ('group', 'batch') and mlp_std is not None:
        mlp_std = 0.04          # un-normalised MLP: keep activations in a sane range
    opt = BaseOptions().parse([])
    opt.mlp_norm = mlp_norm
    opt.norm = norm
    opt.hg_dim, opt.mlp_dim, opt.mlp_res_layers, opt.num_stack = \
        opt.hg_dim_global, opt.mlp_dim_global, opt.mlp_res_layers_global, opt.num_stack_global   # train.py:102-105
    opt.train_full_pifu = False
    netG = PIFuNetwNML(opt, 'orthogonal')
    opt_l = copy.deepcopy(opt)
    opt_l.hg_dim, opt_l.mlp_dim, opt_l.mlp_res_layers, opt_l.num_stack = \
        opt.hg_dim_local, opt.mlp_dim_local, opt.mlp_res_layers_local, opt.num_stack_local       # train.py:115-118
    netMR = PIFuMRNet(opt_l, netG, 'orthogonal')
    netG.eval()                                     # exactly what recon() does (reconstruction.py:288-289,304)
    from pifuhd_amd.config import coarse_opt, fine_opt
    oc = coarse_opt(mlp_norm=mlp_norm, norm=norm)
    of = fine_opt(oc)
    spec = synth.state_dict_spec(oc, of)
    ref_sd = netMR.state_dict()
    assert set(spec) == set(ref_sd), (set(spec) ^ set(ref_sd))
    for k in spec:
        assert tuple(ref_sd[k].shape) == tuple(spec[k]), (k, ref_sd[k].shape, spec[k])
    sd = synth.name_seeded_state_dict(spec, seed=seed, mlp_std=mlp_std, perturb=perturb)
    netMR.load_state_dict(sd, strict=True)
    return netG, netMR, sd, oc, of


def seeded_features(seed, hc=32, hf=64):
    g = torch.Generator().manual_seed(1000 + seed)
    fc = torch.randn(1, 256, hc, hc, generator=g)
    ff = torch.randn(1, 16, hf, hf, generator=g)
    return fc, ff


def seeded_points(seed, n, lim=1.15):
    g = torch.Generator().manual_seed(2000 + seed)
    return (torch.rand(1, 3, n, generator=g) * 2 - 1) * lim


def calib_tensor():
    c = torch.eye(4)
    c[1, 1] = -1.0
    return c[None]


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def golden_query():
    out = {}
    for tag, mlp_norm in (('group', 'group'), ('batch', 'batch'), ('none', 'no')):
        netG, netMR, sd, oc, of = build_reference(mlp_norm=mlp_norm)
        fc, ff = seeded_features(7)
        netG.im_feat_list = [fc]
        netMR.im_feat_list = [ff]
        calib = calib_tensor()
        with torch.no_grad():
            # (a) one chunk of 700 points, some outside the bbox
            pts = seeded_points(3, 700)
            netMR.query(pts, calib)
            out[f'{tag}_a_fine'] = netMR.get_preds()[0, 0].numpy().copy()
            out[f'{tag}_a_coarse'] = netG.preds[0, 0].numpy().copy()
            out[f'{tag}_a_phi_sub'] = netG.phi[0, ::16, :].numpy().copy()
            # (b) 1200 points in chunks of 500 (500, 500, 200) -> chunk-coupled GroupNorm
            pts = seeded_points(4, 1200, lim=1.0)
            preds = []
            for s in range(0, 1200, 500):
                netMR.query(pts[:, :, s:s + 500], calib)
                preds.append(netMR.get_preds()[0, 0].numpy().copy())
            out[f'{tag}_b_fine'] = np.concatenate(preds)
            # (c) coarse-only query (config C1 plumbing)
            pts = seeded_points(5, 300)
            netG.query(pts, calib)
            out[f'{tag}_c_coarse'] = netG.get_preds()[0, 0].numpy().copy()
            # (d) calc_normal on 150 points (600-point GroupNorm batch)
            pts = seeded_points(6, 150, lim=0.9)
            netMR.calc_normal(pts[:, None], calib[:, None], calib)
            out[f'{tag}_d_nml'] = netMR.nmls[0].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'query_small.npz'), **out)
    print('query_small.npz', {k: v.shape for k, v in out.items()})


def golden_filter():
    out = {}
    netG, netMR, sd, oc, of = build_reference()
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        # tiny end-to-end: 64x64 coarse image, 128x128 fine image
        img64 = synth.synthetic_image(64, 3, seed=2)
        netMR.opt.loadSizeBig = 128
        img128 = synth.synthetic_image(128, 3, seed=3)
        netMR.filter_global(img64)
        out['tiny_nmlF'] = netG.nmlF.numpy().copy()
        out['tiny_nmlB'] = netG.nmlB.numpy().copy()
        out['tiny_feat_coarse'] = netG.im_feat_list[0].numpy().copy()           # [1,256,16,16]
        assert len(netG.im_feat_list) == 1
        netMR.filter_local(img128[:, None])                                     # netMR in train mode (C-2)
        assert netMR.training and len(netMR.im_feat_list) == 1
        out['tiny_feat_fine_trainmode'] = netMR.im_feat_list[0].numpy().copy()  # [1,16,64,64]
        # GroupNorm HGFilter variant (real PIFuHD checkpoints use norm='group')
        fg = Filter(1, 2, 9, 16, 'group', 'no_down', False)
        spec = {}
        synth.filter_keys('image_filter', 1, 2, 9, 16, 'group', 'no_down', spec)
        sdg = synth.name_seeded_state_dict(spec, seed=5, perturb=True)
        fg.load_state_dict({k[len('image_filter.'):]: v for k, v in sdg.items()}, strict=True)
        x = torch.randn(1, 9, 64, 64, generator=g)
        out['group_filter_out'] = fg(x)[0][-1].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'filter_small.npz'), **out)
    print('filter_small.npz', {k: v.shape for k, v in out.items()})


def golden_filter_full():
    """Full-size filter_global (512^2) / filter_local (1024^2): strided sub-samples only."""
    out = {}
    netG, netMR, sd, oc, of = build_reference(perturb=False, mlp_std=None, seed=0)
    data = synth.synthetic_inputs(1024, 3, seed=0)
    with torch.no_grad():
        netMR.filter_global(data['img_512'])
        out['nmlF_sub'] = netG.nmlF[0, :, ::16, ::16].numpy().copy()
        out['nmlB_sub'] = netG.nmlB[0, :, ::16, ::16].numpy().copy()
        out['feat_coarse_sub'] = netG.im_feat_list[0][0, ::8, ::8, ::8].numpy().copy()
        netMR.filter_local(data['img'][:, None])
        out['feat_fine_sub'] = netMR.im_feat_list[0][0, :, ::16, ::16].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'filter_full_sub.npz'), **out)
    print('filter_full_sub.npz', {k: v.shape for k, v in out.items()})


def golden_octree():
    """Reference eval_grid / eval_grid_octree bookkeeping on a deterministic noisy analytic field."""
    out = {}
    calib = np.diag([1.0, -1.0, 1.0, 1.0])
    calib_inv = np.linalg.inv(calib)

    def field(points):
        p = points.astype(np.float32).astype(np.float64)
        r = np.sqrt((0.8 * p[0]) ** 2 + (0.45 * p[1]) ** 2 + (1.6 * p[2]) ** 2)
        v = 1.0 / (1.0 + np.exp(-(0.35 - r) / 0.05))
        v = v + 0.02 * np.sin(37.0 * p[0] + 11.0 * p[1]) * np.cos(23.0 * p[2])
        return v.astype(np.float32)

    for R, init in ((32, 4), (64, 8), (64, 16), (64, 64), (48, 6)):
        coords, mat = ref_mesh_util.create_grid(R, R, R)
        coords = coords.reshape(3, -1).T
        coords = np.matmul(np.concatenate([coords, np.ones((coords.shape[0], 1))], 1), calib_inv.T)[:, :3]
        coords = coords.T.reshape(3, R, R, R)
        counts = []

        def counting(points):
            counts.append(points.shape[1])
            return field(points)
        sdf = ref_mesh_util.eval_grid_octree(coords, counting, init_resolution=init, threshold=0.05,
                                             num_samples=5000)
        out[f'oct_{R}_{init}_sha'] = np.frombuffer(hashlib.sha256(sdf.tobytes()).digest(), dtype=np.uint8)
        out[f'oct_{R}_{init}_sub'] = sdf[::3, ::3, ::3].copy()
        out[f'oct_{R}_{init}_chunks'] = np.array(counts, dtype=np.int64)
        out[f'oct_{R}_{init}_sum'] = np.array([sdf.sum()])
        if R == 32:
            dense = ref_mesh_util.eval_grid(coords, field, num_samples=5000)
            out['dense_32_sha'] = np.frombuffer(hashlib.sha256(dense.tobytes()).digest(), dtype=np.uint8)
            out['dense_32_sub'] = dense[::3, ::3, ::3].copy()
    np.savez_compressed(os.path.join(HERE, 'octree_small.npz'), **out)
    print('octree_small.npz', sorted(out))


def golden_recon():
    """Reference `mesh_util.reconstruction` driven end to end with the reference nets; marching
    cubes is captured (SDF saved for make_golden_mc.py) by patching the stubbed skimage entry."""
    out = {}
    netG, netMR, sd, oc, of = build_reference()
    fc, ff = seeded_features(9)
    netG.im_feat_list = [fc]
    netMR.im_feat_list = [ff]
    calib = calib_tensor()
    captured = {}

    def capture(sdf, thresh):
        captured['sdf'] = sdf.copy()
        raise RuntimeError('captured')
    ref_mesh_util.measure.marching_cubes_lewiner = capture
    with torch.no_grad():
        for tag, R, octree in (('dense32', 32, False), ('oct64', 64, True)):
            r = ref_mesh_util.reconstruction(netMR, 'cpu', calib, R, None, None, 0.5, use_octree=octree,
                                             num_samples=5000)
            assert r == -1
            sdf = captured['sdf']
            out[f'{tag}_sdf_f32'] = sdf.astype(np.float32)
            np.save(os.path.join('/tmp', f'golden_sdf_{tag}.npy'), sdf)
    np.savez_compressed(os.path.join(HERE, 'recon_small.npz'), **out)
    print('recon_small.npz', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    which = sys.argv[1:] or ['query', 'filter', 'octree', 'recon', 'filter_full']
    if 'query' in which:
        golden_query()
    if 'filter' in which:
        golden_filter()
    if 'octree' in which:
        golden_octree()
    if 'recon' in which:
        golden_recon()
    if 'filter_full' in which:
        golden_filter_full()
