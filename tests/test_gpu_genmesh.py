"""End-to-end gen_mesh (reconstruction.py:25-72) on the GPU against the oracle pipeline, the coarse-only
configuration (BASELINE config 1) and the sharded evaluation entry points on a single rank."""
import os

import numpy as np
import pytest
import torch

from helpers import calib_tensor, golden_nets, seeded_features

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('configuration', ['default', 'pifuhd_pt'])
def test_gen_mesh_end_to_end(tmp_path, configuration):
    """`configuration` 'pifuhd_pt': GroupNorm filters and an MLP without norm layers - what the published checkpoint's opt
    selects (`reconstruction.py:260-268`, `MLP.py:37-41`)."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import nets, reconstruction, synth
    sd, oc, of = golden_nets() if configuration == 'default' else golden_nets(mlp_norm='no', norm='group', mlp_std=0.10)
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = {'name': 't', 'img': synth.synthetic_image(128, 3, seed=3), 'img_512': synth.synthetic_image(64, 3, seed=2),
            'calib': calib_tensor(), 'b_min': [-1, -1, -1], 'b_max': [1, 1, 1]}
    # make sure the 0.5 level set exists for these random weights
    netMR.filter_global(data['img_512']); netMR.filter_local(data['img'][:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3, resolution=32)
    path = str(tmp_path / 'mesh.obj')
    timings = {}
    # like the reference, an octree grid below init_resolution = 64 evaluates nothing, marching cubes fails,
    # reconstruction() returns -1 and the tuple-unpack raises TypeError (mesh_util.py:138-140,94-96; reconstruction.py:56)
    with pytest.raises(TypeError):
        reconstruction.gen_mesh(48, netMR, 'cuda', data, path)
    verts, faces, color = reconstruction.gen_mesh(64, netMR, 'cuda', data, path, timings=timings)
    assert os.path.exists(path) and os.path.exists(path[:-4] + '.png')
    assert len(verts) > 100 and len(faces) > 100 and timings['n_verts'] == len(verts)
    # reference quirk (SURVEY C-6): the last vertex is never coloured
    assert np.all(color[-1] == 0) and np.abs(color[:-1]).sum() > 0
    # OBJ text = the reference's format applied to the returned arrays
    assert open(path).read() == O.obj_text(verts, faces, color)
    # colours = calc_normal * 0.5 + 0.5 of the oracle on the same vertices (forward differences: compare by angle)
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    with torch.no_grad():
        ref.filter_global(data['img_512']); ref.filter_local(data['img'])
        vt = torch.from_numpy(verts.T).float()[None]
        n_ref = ref.calc_normal(vt[:, :, :-1], data['calib'])[0].numpy()
    n_got = (color[:-1] - 0.5) / 0.5
    cos = (n_got.T * n_ref).sum(0) / np.maximum(np.linalg.norm(n_got.T, axis=0), 1e-9)
    assert np.median(cos) > 0.999 and (cos > 0.98).mean() > 0.9
    # mesh = marching cubes of the device SDF: faces bit-exact vs the serial oracle on the same field
    from pifuhd_amd import mesh_util
    res = mesh_util.reconstruction(netMR, 'cuda', data['calib'], 64, None, None, 0.5, use_octree=True, num_samples=5000,
                                   return_sdf=True)
    v2, f2, _, _, sdf = res
    assert np.array_equal(f2, faces) and np.array_equal(v2, verts)
    vr, fr = O.marching_cubes_lewiner(sdf.cpu().numpy(), 0.5)
    assert np.array_equal(faces[:, ::-1], fr)
    # and the SDF itself agrees with the oracle's octree evaluation of the oracle network
    def eval_func(points):
        with torch.no_grad():
            return ref.query(torch.from_numpy(points[None]).float(), data['calib'])[0, 0].numpy()
    calib_inv = np.linalg.inv(data['calib'][0].numpy().astype(np.float64))
    sdf_ref = O.eval_grid_octree(64, calib_inv, eval_func, 5000)
    assert np.abs(sdf.cpu().numpy() - sdf_ref).max() < 1e-4


def test_coarse_only_reconstruction_config1():
    """BASELINE config 1: PIFuNetwNML only (opt.loadSize = 512 => depth scale 1.28), dense grid through
    mesh_util.reconstruction (SURVEY C-9: gen_mesh itself needs the fine net's calc_normal)."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import config, mesh_util, nets, synth
    oc = config.coarse_opt(loadSize=512)
    spec = {}
    synth.mlp_keys('mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
    sd = synth.name_seeded_state_dict(spec, seed=11, mlp_std=0.16, perturb=True)
    netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda')
    netG.eval()
    netG._sd = {k: v.cuda() for k, v in sd.items()}
    netG._push_weights()
    fc, _ = seeded_features(4)
    netG.set_features(fc)
    calib = calib_tensor()
    sdf, n = mesh_util.eval_grid_device(netG, calib, 32, 5000, use_octree=False)
    assert n == 32 ** 3
    ref = O.OracleNets({'netG.' + k: v for k, v in sd.items()}, oc)
    ref.feat_coarse = fc

    def eval_func(points):
        with torch.no_grad():
            return ref.query_coarse(torch.from_numpy(points[None]).float(), calib)[0][0, 0].numpy()
    want = O.eval_grid(32, np.linalg.inv(calib[0].numpy().astype(np.float64)), eval_func, 5000)
    assert np.abs(sdf.cpu().numpy() - want).max() < 1e-4


def test_sharded_entry_points_single_rank():
    """dist.eval_grid_octree_sharded / eval_lattice_sharded on one rank must reproduce pifu_eval_grid bit for bit."""
    from pifuhd_amd import dist as pdist, mesh_util, nets
    sd, oc, of = golden_nets()
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(9)
    netG.set_features(fc); netMR.set_features(ff)
    calib = calib_tensor()
    a, na = mesh_util.eval_grid_device(netMR, calib, 64, 5000, use_octree=True, init_resolution=16)
    b, nb = pdist.eval_grid_octree_sharded(netMR, calib, 64, 5000, init_resolution=16)
    assert na == nb and torch.equal(a, b)
    dense = pdist.eval_lattice_sharded(netMR, calib, (16, 32, 24), 5000)
    d2, _ = mesh_util.eval_grid_device(netMR, calib, 32, 5000, use_octree=False)
    assert dense.shape[0] == 16 * 32 * 24 and torch.isfinite(dense).all() and d2.shape == (32, 32, 32)


def test_gen_mesh_imgcolor(tmp_path, capsys):
    """`gen_mesh_imgColor` (`reconstruction.py:77-125`): 100 000-point chunks, colours = index(image, projected verts),
    vertices through inv(calib_world); failures are printed, not raised."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import mesh_util, nets, reconstruction, synth
    sd, oc, of = golden_nets()
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    world = torch.tensor([[[1.7, 0, 0, 0.2], [0, 1.7, 0, -0.1], [0, 0, 1.7, 0], [0, 0, 0, 1]]])
    data = {'name': 't', 'img': synth.synthetic_image(128, 3, seed=3), 'img_512': synth.synthetic_image(64, 3, seed=2),
            'calib': calib_tensor(), 'calib_world': world, 'b_min': [-1, -1, -1], 'b_max': [1, 1, 1]}
    netMR.filter_global(data['img_512']); netMR.filter_local(data['img'][:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3, resolution=32)
    path = str(tmp_path / 'mesh.obj')
    out = reconstruction.gen_mesh_imgColor(64, netMR, 'cuda', data, path)
    assert out is not None and os.path.exists(path) and os.path.exists(path[:-4] + '.png')
    verts, faces, color = out
    # the mesh before the world transform = reconstruction() with 100 000-point chunks
    v0, f0, _, _ = mesh_util.reconstruction(netMR, 'cuda', data['calib'], 64, None, None, 0.5, use_octree=True,
                                            num_samples=100000)
    assert np.array_equal(f0, faces)
    want_v = np.matmul(np.concatenate([v0, np.ones_like(v0[:, :1])], 1), np.linalg.inv(world[0].numpy()).T)[:, :3]
    assert np.array_equal(want_v, verts)
    xyz = O.orthogonal(torch.from_numpy(v0.T).float()[None], data['calib'])
    want_c = O.index(data['img'][:1], xyz[:, :2])[0].numpy().T * 0.5 + 0.5
    assert np.abs(color - want_c).max() < 1e-5
    assert open(path).read() == O.obj_text(verts, faces, color)
    # chunk size is part of the numerical contract: 5 000-point chunks give a different field
    v5, f5, _, _ = mesh_util.reconstruction(netMR, 'cuda', data['calib'], 64, None, None, 0.5, use_octree=True, num_samples=5000)
    assert v5.shape != v0.shape or not np.array_equal(v5, v0)
    # errors are swallowed and printed, like the reference's try/except
    capsys.readouterr()
    assert reconstruction.gen_mesh_imgColor(48, netMR, 'cuda', data, path) is None
    assert 'error' in capsys.readouterr().out


def test_recon_driver_checkpoint_roundtrip(tmp_path, capsys):
    """`recon` (`reconstruction.py:231-321`): checkpoint {'opt','opt_netG','model_state_dict'} -> nets -> dataset loop."""
    import argparse
    from PIL import Image
    from pifuhd_amd import nets, reconstruction, synth
    sd, oc, of = golden_nets()
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    # a subject on disk: 128 px PNG + rect covering it
    img = ((synth.synthetic_image(128, 3, seed=3)[0].permute(1, 2, 0).numpy() * 0.5 + 0.5) * 255).round().astype(np.uint8)
    root = tmp_path / 'data'; os.makedirs(root)
    Image.fromarray(img).save(root / 'subj.png')
    (root / 'subj_rect.txt').write_text('0 0 128 128\n')
    from pifuhd_amd.readData import readDataset
    ds = readDataset(str(root), loadsize=128)
    data = ds[0]
    netMR.filter_global(data['img_512']); netMR.filter_local(data['img'][:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3, resolution=32)
    opt_ck = argparse.Namespace(**vars(of)); opt_ck.name = 'unit'; opt_ck.checkpoints_path = str(tmp_path / 'ck')
    opt_ck.use_compose = False
    os.makedirs(opt_ck.checkpoints_path)
    ck = os.path.join(opt_ck.checkpoints_path, 'unit_train_latest')
    reconstruction.save_checkpoint(ck, opt_ck, oc, netMR)
    loaded = reconstruction.load_checkpoint(ck)
    assert set(loaded) == {'opt', 'opt_netG', 'model_state_dict'} and loaded['opt'].name == 'unit'
    assert set(loaded['model_state_dict']) == set(netMR.state_dict())
    want_v, want_f, want_c = reconstruction.gen_mesh(64, netMR, 'cuda', data, str(tmp_path / 'direct.obj'))
    opt = argparse.Namespace(load_netMR_checkpoint_path=None, resume_epoch=-1, checkpoints_path=opt_ck.checkpoints_path,
                             name='unit', gpu_id=0, start_id=-1, end_id=-1, use_color=0, dataroot=str(root), resolution=64,
                             results_path=str(tmp_path / 'results'), loadSize=128)
    written = reconstruction.recon(opt, use_crop=False, use_gpu=True)
    assert written == [str(tmp_path / 'results') + '/unit/recon/result_subj_64.obj']
    assert open(written[0]).read() == open(tmp_path / 'direct.obj').read()
    with pytest.raises(Exception, match='Failed loading state dict'):
        opt.name = 'nope'; opt.resume_epoch = 3
        reconstruction.recon(opt)
    # loadFromPIFu: same architecture => everything copied; a wider MLP keeps its own mlp tensors
    sd2, _, _ = golden_nets()
    other = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda'); other.eval()
    other.load_state_dict({k: v + 1.0 if v.dtype.is_floating_point else v for k, v in sd2.items()}, _prefix='netG.')
    capsys.readouterr()
    other.loadFromPIFu(netG)
    out = capsys.readouterr().out
    assert 'not initialized []' in out and 'mlp not initialized []' in out
    assert all(torch.equal(other.state_dict()[k].cpu(), netG.state_dict()[k].cpu()) for k in netG.state_dict()
               if k.startswith(('image_filter.', 'mlp.')))


def test_esti_color_bit_exact_against_the_reference_function(golden_dir):
    """`pifu_esti_color` on the fixture produced by executing the REFERENCE's `esti_color` (ties on lattice coordinates,
    short candidate lists, both tie directions), and on a larger seeded case against the oracle."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import reconstruction
    gold = np.load(os.path.join(golden_dir, 'esti_color.npz'))
    for i in range(int(gold['n_cases'][0])):
        xyz = torch.from_numpy(gold[f'c{i}_xyz'].T.copy())[None].cuda()
        got = reconstruction.esti_color(gold[f'c{i}_color_in'], xyz)
        assert np.array_equal(got, gold[f'c{i}_color_out']), i
    rs = np.random.RandomState(11)
    n = 6000
    xyz = np.concatenate([np.round(rs.rand(n, 2) * 40) / 20 - 1, (rs.rand(n, 1) - 0.5) * np.where(rs.rand(n, 1) < 0.2, 0.002, 1.0)], 1).astype(np.float32)
    xyz[rs.rand(n) < 0.1, 2] = 0.0005
    col = rs.randint(0, 256, size=(n, 4)).astype(np.uint16)
    got = reconstruction.esti_color(col, torch.from_numpy(xyz.T.copy())[None].cuda())
    assert np.array_equal(got, O.esti_color(col, xyz))
    with pytest.raises(Exception):
        reconstruction.esti_color(np.zeros((2, 4), np.uint16), torch.tensor([[[0.0, 0.5], [0.0, 0.5], [-0.1, 0.5]]]).cuda())


def test_face_components_against_the_oracle():
    """Device union-find over faces: labels (= smallest face index of the component) and open-edge flags equal the serial
    oracle's on a marching-cubes mesh of a noise volume (hundreds of closed blobs, sheets cut by the volume border)."""
    import ctypes as C
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib, mesh_util
    from pifuhd_amd.nets import _Ctx, _ptr, _stream_ptr
    ctx = _Ctx('cuda')
    rs = np.random.RandomState(3)
    vol = rs.rand(28, 30, 26).astype(np.float32)
    for ax in range(3):
        vol = (np.roll(vol, 1, ax) + vol + np.roll(vol, -1, ax)) / 3.0
    v, f = mesh_util.marching_cubes_device(ctx, torch.from_numpy(vol).cuda(), float(np.median(vol)) + 0.02)
    faces = f.cpu().numpy()
    label = torch.empty(len(faces), device='cuda', dtype=torch.int32)
    open_edge = torch.empty(len(faces), device='cuda', dtype=torch.uint8)
    _lib.call('pifu_face_components', ctx.h, _ptr(f.contiguous()), len(faces), _ptr(label), _ptr(open_edge), _stream_ptr(ctx.device))
    want_label, want_open = O.face_components(faces)
    assert len(np.unique(want_label)) > 5
    assert np.array_equal(label.cpu().numpy(), want_label)
    assert np.array_equal(open_edge.cpu().numpy(), want_open)


def _plus_oracle(base_obj, calib):
    """gen_mesh_imgColor_plus after the image-coloured OBJ exists, restated with the oracle: returns (verts, faces, rgba)
    or the exception the pipeline ends with."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import reconstruction
    v0, c0, f0 = reconstruction.read_obj_with_color(base_obj)
    v0, c0, f0 = O.merge_vertices(v0, c0, f0)                     # trimesh.load(process=True)
    try:
        keep = O.widest_watertight_component(v0, f0)               # mesh.split() + the extent loop
    except IndexError as e:
        return e
    used = np.unique(f0[keep])
    remap = np.full(len(v0), -1)
    remap[used] = np.arange(len(used))
    verts, faces = v0[used], remap[f0[keep]]
    rgba0 = np.concatenate([np.clip(np.rint(c0[used] * 255.0), 0, 255), np.full((len(used), 1), 255.0)], 1).astype(np.uint16)
    xyz = O.orthogonal(torch.from_numpy(verts.T).float()[None], calib)[0].numpy().T
    try:
        return verts, faces, O.esti_color(rgba0, xyz).astype(np.uint8)
    except Exception as e:
        return e


def test_meshcleaning_and_recolouring_on_analytic_bodies(tmp_path):
    """meshcleaning + esti_color on a mesh with known structure: a wide closed ellipsoid, a taller but narrower closed one,
    and a sheet cut open by the volume border (wider than both, but not watertight: trimesh's split() drops it)."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import mesh_util, reconstruction
    from pifuhd_amd.nets import _Ctx
    R = 64
    lin = -1 + 2 * np.arange(R) / R
    X, Y, Z = np.meshgrid(lin, lin, lin, indexing='ij')
    wide = 1 - np.sqrt(((X + 0.1) / 0.55) ** 2 + ((Y - 0.45) / 0.2) ** 2 + (Z / 0.3) ** 2)
    tall = 1 - np.sqrt(((X - 0.3) / 0.2) ** 2 + ((Y + 0.35) / 0.4) ** 2 + (Z / 0.25) ** 2)
    sheet = 0.03 - np.abs(Y + 0.85)
    vol = (0.5 + 0.5 * np.maximum(np.maximum(wide, tall), sheet * 10)).astype(np.float32)
    v, f = mesh_util.marching_cubes_device(_Ctx('cuda'), torch.from_numpy(vol).cuda(), 0.5)
    verts = v.cpu().numpy().astype(np.float64) * (2.0 / R) - 1.0
    faces = f.cpu().numpy()
    rs = np.random.RandomState(2)
    colour = rs.rand(len(verts), 3)
    base = str(tmp_path / 'bodies.obj')
    mesh_util.save_obj_mesh_with_color(base, verts, faces, colour)
    want = _plus_oracle(base, calib_tensor())
    assert not isinstance(want, Exception)
    kept_v, kept_rgba, kept_f = reconstruction.meshcleaning(base, 'cuda')
    assert np.array_equal(kept_v, want[0]) and np.array_equal(kept_f, want[1])
    assert kept_v[:, 1].min() > 0.2 and np.ptp(kept_v[:, 0]) > 1.0              # the wide ellipsoid, not the tall one, not the sheet
    xyz = O.orthogonal(torch.from_numpy(kept_v.T).float()[None], calib_tensor())
    got = reconstruction.esti_color(kept_rgba.astype(np.uint16), xyz.cuda()).astype(np.uint8)
    assert np.array_equal(got, want[2])
    assert (got != kept_rgba).any(1).sum() > 100                                  # the back half was recoloured
    # the rewritten file holds exactly the kept component
    v1, c1, f1 = reconstruction.read_obj_with_color(base)
    assert np.array_equal(f1, kept_f) and np.allclose(v1, kept_v, atol=5e-5)
    # only open components: the reference's cc[0] raises IndexError
    only_sheet = str(tmp_path / 'sheet.obj')
    v, f = mesh_util.marching_cubes_device(_Ctx('cuda'), torch.from_numpy((0.5 + 5 * sheet).astype(np.float32)).cuda(), 0.5)
    mesh_util.save_obj_mesh_with_color(only_sheet, v.cpu().numpy().astype(np.float64), f.cpu().numpy(), np.zeros((len(v), 3)))
    with pytest.raises(IndexError):
        reconstruction.meshcleaning(only_sheet, 'cuda')


def test_meshcleaning_merges_coincident_vertices_like_the_trimesh_loader(tmp_path):
    """`trimesh.load` merges vertices that share a position before `split()` runs, and the OBJ carries 4 decimals: the
    marching-cubes vertices around a grid node whose value all but equals the level coincide after the round trip.  The
    merge turns the sliver triangles between them into degenerate ones (kept: validate=False); each has an edge no other
    face uses, so it is a non-watertight component of its own and `split()` drops it, while the body around it closes up
    over the merged vertex (its faces on either side of a sliver now share an edge) and stays watertight.  The product
    must do exactly what the restated pipeline of the oracle does: same kept vertices (fewer than the file holds, in file
    order), same faces."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import mesh_util, reconstruction
    from pifuhd_amd.nets import _Ctx
    R = 64
    lin = -1 + 2 * np.arange(R) / R
    X, Y, Z = np.meshgrid(lin, lin, lin, indexing='ij')
    wide = 1 - np.sqrt(((X + 0.1) / 0.55) ** 2 + ((Y - 0.45) / 0.2) ** 2 + (Z / 0.3) ** 2)
    tall = 1 - np.sqrt(((X - 0.3) / 0.2) ** 2 + ((Y + 0.35) / 0.4) ** 2 + (Z / 0.25) ** 2)
    vol = (0.5 + 0.5 * np.maximum(wide, tall)).astype(np.float32)
    # an inside node of the wide body with two or more outside neighbours gets a value a hair above the level: the crossings
    # on its edges land within 1e-5 cells of it.  Even indices: the node's coordinates (multiples of 1/16) are exact at 4
    # decimals, so the crossings round onto the node from either side (odd nodes sit on rounding ties)
    inside = vol > 0.5
    outside_nb = sum((~np.roll(inside, sh, axis=ax)).astype(int) for ax in range(3) for sh in (1, -1))
    I, J, K = np.indices(vol.shape)
    cand = np.argwhere(inside & (outside_nb >= 2) & (Y > 0.2) & (I % 2 == 0) & (J % 2 == 0) & (K % 2 == 0))
    node = tuple(cand[len(cand) // 2])
    vol[node] = np.float32(0.5 + 2e-7)
    v, f = mesh_util.marching_cubes_device(_Ctx('cuda'), torch.from_numpy(vol).cuda(), 0.5)
    verts = v.cpu().numpy().astype(np.float64) * (2.0 / R) - 1.0
    faces = f.cpu().numpy()
    colour = np.random.RandomState(5).rand(len(verts), 3)
    base = str(tmp_path / 'coincident.obj')
    mesh_util.save_obj_mesh_with_color(base, verts, faces, colour)
    v0, c0, f0 = reconstruction.read_obj_with_color(base)
    assert len(np.unique(v0, axis=0)) < len(v0)                                   # the file really holds coincident vertices
    mv, mc, mf = reconstruction.merge_vertices(v0, c0, f0)
    ov, oc_, of_ = O.merge_vertices(v0, c0, f0)
    assert np.array_equal(mv, ov) and np.array_equal(mc, oc_) and np.array_equal(mf, of_)
    assert len(mv) < len(v0) and len(mf) == len(f0)                               # merged; no face dropped
    degenerate = (mf[:, 0] == mf[:, 1]) | (mf[:, 1] == mf[:, 2]) | (mf[:, 0] == mf[:, 2])
    assert degenerate.any()                                                       # the slivers stay in the face list
    # the component test itself against the serial restatement, on the merged mesh
    keep = reconstruction.largest_component(mv, mf, 'cuda')
    assert np.array_equal(keep, O.widest_watertight_component(mv, mf))
    assert not degenerate[keep].any()                                             # ... but in no watertight component
    want = _plus_oracle(base, calib_tensor())
    assert not isinstance(want, Exception)
    kept_v, kept_rgba, kept_f = reconstruction.meshcleaning(base, 'cuda')
    assert np.array_equal(kept_v, want[0]) and np.array_equal(kept_f, want[1])
    assert kept_v[:, 1].min() > 0.2 and np.ptp(kept_v[:, 0]) > 1.0                # still the wide body, closed over the merged vertex
    assert len(np.unique(kept_v, axis=0)) == len(kept_v)


def test_gen_mesh_imgcolor_plus(tmp_path, capsys):
    """`gen_mesh_imgColor_plus` (`reconstruction.py:127-178`) end to end on the network's own mesh: the result - a cleaned,
    recoloured mesh or, exactly like the reference, a printed error when the widest watertight component has no silhouette
    vertex - must be what the oracle pipeline gives on the same image-coloured OBJ."""
    from pifuhd_amd import nets, reconstruction, synth
    sd, oc, of = golden_nets()
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = {'name': 't', 'img': synth.synthetic_image(128, 3, seed=3), 'img_512': synth.synthetic_image(64, 3, seed=2),
            'calib': calib_tensor(), 'b_min': [-1, -1, -1], 'b_max': [1, 1, 1]}
    netMR.filter_global(data['img_512']); netMR.filter_local(data['img'][:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3, resolution=32)
    base = str(tmp_path / 'base.obj')
    assert reconstruction.gen_mesh_imgColor(64, netMR, 'cuda', data, base) is not None
    want = _plus_oracle(base, data['calib'])
    path = str(tmp_path / 'plus.obj')
    capsys.readouterr()
    out = reconstruction.gen_mesh_imgColor_plus(64, netMR, 'cuda', data, path)
    shown = capsys.readouterr().out
    assert 'Processing mesh cleaning' in shown
    if isinstance(want, Exception):
        assert out is None and 'error' in shown
        return
    verts, faces, rgba = out
    assert np.array_equal(verts, want[0]) and np.array_equal(faces, want[1]) and np.array_equal(rgba, want[2])
    v1, c1, f1 = reconstruction.read_obj_with_color(path)
    assert np.array_equal(f1, faces) and np.allclose(v1, verts, atol=5e-5) and np.allclose(c1, rgba[:, :3] / 255.0, atol=5e-5)


def test_result_arrays_are_the_callers_own():
    """`reconstruction()` hands its vertex / face arrays out of pinned slabs that it reuses (`mesh_util.to_host`): a result
    the caller still holds - directly or through a view - must never be overwritten by a later call, and memory is taken
    back only once the caller has dropped the arrays."""
    import gc
    from pifuhd_amd import mesh_util, nets
    sd, oc, of = golden_nets()
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(9)
    netG.set_features(fc)
    netMR.set_features(ff)

    def run(R):
        r = mesh_util.reconstruction(netMR, 'cuda', calib_tensor(), R, None, None, 0.5, use_octree=True, num_samples=5000)
        assert r != -1
        return r[0], r[1]
    va, fa = run(128)
    assert va.nbytes > (4 << 20)                                   # large enough to come out of a slab
    keep_v, keep_f = va.copy(), fa.copy()
    view_only = fa[:, ::-1]                                        # a view alone must keep the memory, too
    addr = {va.ctypes.data, fa.ctypes.data}
    del fa
    vb, fb = run(128)                                              # while the first result is still referenced
    assert not ({vb.ctypes.data, fb.ctypes.data} & addr)
    assert np.array_equal(va, keep_v) and np.array_equal(view_only[:, ::-1], keep_f)
    assert np.array_equal(vb, keep_v) and np.array_equal(fb, keep_f)
    vc, fc_ = run(96)                                              # a third, different result: still nothing is clobbered
    assert np.array_equal(va, keep_v) and np.array_equal(vb, keep_v) and np.array_equal(view_only[:, ::-1], keep_f)
    del va, view_only, vb, fb, vc, fc_
    gc.collect()
    vd, fd = run(128)                                              # everything dropped: the slabs are taken back
    assert {vd.ctypes.data, fd.ctypes.data} & addr
    assert np.array_equal(vd, keep_v) and np.array_equal(fd, keep_f)


def test_gen_mesh_is_reproducible_run_to_run(tmp_path):
    """The marching-cubes surface list is filled in a run-dependent order (one atomic per block) and the OBJ file is written
    by many host threads at offsets derived ahead of time: neither may show in the results.  Four runs at 128^3 (several
    counting blocks, several writer blocks): identical arrays and identical file bytes."""
    import hashlib
    from pifuhd_amd import nets, reconstruction, synth
    sd, oc, of = golden_nets()
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = {'name': 't', 'img': synth.synthetic_image(128, 3, seed=3), 'img_512': synth.synthetic_image(64, 3, seed=2),
            'calib': calib_tensor(), 'b_min': [-1, -1, -1], 'b_max': [1, 1, 1]}
    netMR.filter_global(data['img_512']); netMR.filter_local(data['img'][:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3, resolution=32)
    seen = set()
    for k in range(4):
        p = str(tmp_path / f'r{k}.obj')
        v, f, c = reconstruction.gen_mesh(128, netMR, 'cuda', data, p)
        h = hashlib.sha256()
        for a in (v, f, c):
            h.update(np.ascontiguousarray(a).tobytes())
        seen.add((h.hexdigest(), hashlib.sha256(open(p, 'rb').read()).hexdigest()))
        assert len(v) > 20000
    assert len(seen) == 1


def test_result_pool_is_bounded_and_can_be_released():
    """The pinned result pool (`mesh_util.to_host`) keeps at most POOL_MAX_BYTES page-locked, gives back a free slab that is
    far larger than what is asked for, and `release_pinned()` returns everything that is not referenced any more."""
    import gc
    from pifuhd_amd import mesh_util
    mesh_util.release_pinned()
    saved = mesh_util.POOL_MAX_BYTES
    try:
        big = torch.arange(48 << 20, dtype=torch.uint8, device='cuda')               # 48 MiB
        a, a2, a3 = mesh_util.to_host(big), mesh_util.to_host(big[:40 << 20]), mesh_util.to_host(big[:44 << 20])
        assert np.array_equal(a[:1000], np.arange(1000, dtype=np.uint8)) and mesh_util._pool_bytes() >= (132 << 20)
        assert len(mesh_util._slabs) == 3
        del a, a2, a3
        gc.collect()
        small = torch.arange(5 << 20, dtype=torch.uint8, device='cuda')              # 5 MiB: the free 50-60 MiB slabs are > 4x that
        b = mesh_util.to_host(small)
        assert mesh_util._pool_bytes() < (16 << 20) and len(mesh_util._slabs) == 1
        mesh_util.POOL_MAX_BYTES = mesh_util._pool_bytes() + (1 << 20)
        c = mesh_util.to_host(big)                                                    # over the cap: a plain pageable copy
        assert mesh_util._pool_bytes() < (16 << 20) and np.array_equal(c[-3:], big[-3:].cpu().numpy())
        held = mesh_util._pool_bytes()
        assert mesh_util.release_pinned() == 0 or mesh_util._pool_bytes() == held     # b is still referenced: its slab stays
        assert np.array_equal(b[:7], np.arange(7, dtype=np.uint8))
        del b, c
        gc.collect()
        assert mesh_util.release_pinned() >= held and mesh_util._pool_bytes() == 0
    finally:
        mesh_util.POOL_MAX_BYTES = saved


def test_result_pool_makes_room_for_growing_results():
    """Subjects of growing size: once POOL_MAX_SLABS slabs exist and all of them are free but too small, the next result must
    not fall back to the pageable, synchronous copy for the rest of the process (the round-4 advisor's finding) - the pool
    drops its smallest free slabs and allocates one that fits, so `to_host_async` keeps returning an event."""
    import gc
    from pifuhd_amd import mesh_util
    mesh_util.release_pinned()
    src = torch.arange(96 << 20, dtype=torch.uint8, device='cuda')
    sizes = [6 << 20, 8 << 20, 10 << 20, 12 << 20, 14 << 20, 16 << 20]                # six slabs, all in use at once
    held = [mesh_util.to_host(src[:n]) for n in sizes]
    assert len(mesh_util._slabs) == mesh_util.POOL_MAX_SLABS == 6
    del held
    gc.collect()
    # every slab is free, none holds 40 MiB, none is more than four times too large for it to be given back on sight
    arr, ev = mesh_util.to_host_async(src[:40 << 20])
    assert ev is not None, 'fell back to the pageable copy although free slabs could make room'
    ev.synchronize()
    assert np.array_equal(arr[-5:], src[(40 << 20) - 5:40 << 20].cpu().numpy())
    assert len(mesh_util._slabs) <= mesh_util.POOL_MAX_SLABS and any(s[0].numel() >= (40 << 20) for s in mesh_util._slabs)
    # a slab still in use is never dropped to make room
    arr2, ev2 = mesh_util.to_host_async(src[:80 << 20])
    assert ev2 is not None
    ev2.synchronize()
    assert np.array_equal(arr[:9], np.arange(9, dtype=np.uint8)) and np.array_equal(arr2[:9], np.arange(9, dtype=np.uint8))
    del arr, arr2
    gc.collect()
    mesh_util.release_pinned()


def test_gen_mesh_failure_leaves_no_png_thread_behind(tmp_path):
    """gen_mesh starts the PNG encoder on a thread that reads a pinned buffer of this (device, thread).  When the call fails
    after that - here marching cubes finds no surface, reconstruction() returns -1 and the tuple-unpack raises TypeError like
    the reference (`reconstruction.py:56`) - the thread must have been joined before the exception leaves gen_mesh, so the
    next subject cannot overwrite pixels that are still being encoded; scratch buffers are per (tag, device, thread)."""
    import threading
    from pifuhd_amd import mesh_util, nets, reconstruction, synth
    sd, oc, of = golden_nets()
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = {'name': 't', 'img': synth.synthetic_image(128, 3, seed=3), 'img_512': synth.synthetic_image(64, 3, seed=2),
            'calib': calib_tensor(), 'b_min': [-1, -1, -1], 'b_max': [1, 1, 1]}
    before = set(threading.enumerate())
    for k in range(2):
        path = str(tmp_path / f'fail{k}.obj')
        with pytest.raises(TypeError):
            reconstruction.gen_mesh(48, netMR, 'cuda', data, path)        # octree grid below init_resolution: no surface
        assert not [t for t in threading.enumerate() if t not in before and t.is_alive()]
        assert os.path.getsize(path[:-4] + '.png') > 0 and not os.path.exists(path)
    keys = [k for k in mesh_util._pinned if k[0] == 'png']
    assert keys and all(k[1] == torch.device('cuda', torch.cuda.current_device()).index and k[2] == threading.get_ident() for k in keys)
    other = {}
    th = threading.Thread(target=lambda: other.update(buf=mesh_util.pinned_flat('png', 16, torch.float32, 'cuda:0')))
    th.start(); th.join()
    assert all(other['buf'].data_ptr() != mesh_util._pinned[k].data_ptr() for k in keys)


@pytest.mark.timeout(900)
def test_config3_network_octree_512_gen_mesh_as_the_metric_runs_it(tmp_path, observed):
    """BASELINE config 3 exactly as bench.py times it (the second half of the metric): `gen_mesh(512, ...)` with the
    two-level network on the bench's subject - RGB-D input, full-size encoders, name-seeded weights, octree refinement from
    64^3 (`reconstruction.py:25-72`, `mesh_util.py:124-187`).  Too large for the CPU oracle as a whole, so it is pinned piece
    by piece: (1) two calls give the same arrays and the same OBJ bytes; (2) the fused device loop and the level-by-level ABI
    calls select the same number of points on every level and leave the identical float64 SDF; (3) on every level at least
    three whole 5 000-point GroupNorm chunks - the first, one from the middle, the ragged last - agree with the CPU oracle
    within the north star's 1e-4; (4) marching cubes on a 128^3 crop of the SDF around the surface is the serial oracle's
    (`oracle/mc_lewiner.c`), vertices and faces bit for bit; (5) the returned mesh is the marching cubes of that SDF."""
    import ctypes as C
    import filecmp
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib, config, mesh_util, nets, reconstruction, synth
    from pifuhd_amd.nets import _ptr, _stream_ptr
    R, chunk = 512, 5000
    oc = config.coarse_opt(in_image_channels=6)
    of = config.fine_opt(oc)
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = synth.synthetic_inputs(1024, 6, seed=0)
    netMR.filter_global(data['img_512'].cuda())
    netMR.filter_local(data['img'].cuda()[:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3)
    calib_t = data['calib']

    # (1) the call itself, twice
    t1, t2 = {}, {}
    a = reconstruction.gen_mesh(R, netMR, 'cuda', data, str(tmp_path / 'a.obj'), timings=t1)
    a = tuple(np.array(x) for x in a)                      # the arrays are views of pinned slabs: keep copies
    b = reconstruction.gen_mesh(R, netMR, 'cuda', data, str(tmp_path / 'b.obj'), timings=t2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert filecmp.cmp(tmp_path / 'a.obj', tmp_path / 'b.obj', shallow=False)
    verts, faces, color = a
    assert len(verts) > 1_000_000 and len(faces) > 2_000_000 and np.all(color[-1] == 0)
    assert faces.min() == 0 and faces.max() == len(verts) - 1
    del b

    # (2) + (3): the octree level by level through the public entry points, sampled chunks against the oracle
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    ref.feat_coarse = netG.im_feat_list[-1].cpu()
    ref.feat_fine = netMR.im_feat_list[-1].cpu()
    calib, cinv = mesh_util.calib_pair(calib_t)
    ctx = netMR._ctx
    st = _stream_ptr(ctx.device)
    res_a = (C.c_int32 * 3)(R, R, R)
    sdf2 = torch.empty(R, R, R, device='cuda', dtype=torch.float64)
    notproc = torch.empty(R, R, R, device='cuda', dtype=torch.uint8)
    _lib.call('pifu_octree_begin', ctx.h, R, _ptr(sdf2), _ptr(notproc), st)
    reso, levels, worst = R // 64, [], {}
    while reso > 0:
        cnt = C.c_int64(0)
        _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), None, 0, C.byref(cnt), st)
        n = cnt.value
        levels.append((reso, n))
        idx = torch.empty(n, device='cuda', dtype=torch.int64)
        vals = torch.empty(n, device='cuda', dtype=torch.float32)
        _lib.call('pifu_octree_select', ctx.h, R, reso, _ptr(notproc), _ptr(idx), n, C.byref(cnt), st)
        _lib.call('pifu_eval_lattice', ctx.h, res_a, cinv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p),
                  _ptr(idx), 0, n, chunk, 0, _ptr(vals), st)
        nchunks = -(-n // chunk)
        assert nchunks >= 3
        err = 0.0
        for c in sorted({0, nchunks // 2, nchunks - 1}):
            lo, hi = c * chunk, min((c + 1) * chunk, n)
            pts = mesh_util.lattice_points(R, cinv, idx[lo:hi].cpu().numpy()).astype(np.float32)
            with torch.no_grad():
                want = ref.query(torch.from_numpy(pts)[None], calib_t)[0, 0].numpy()
            err = max(err, float(np.abs(vals[lo:hi].cpu().numpy() - want).max()))
        worst['reso_%d' % reso] = err
        assert err < 1e-4, (reso, err)
        _lib.call('pifu_octree_commit', ctx.h, R, reso, 0.05, _ptr(idx), _ptr(vals), n, _ptr(sdf2), _ptr(notproc), st)
        if reso <= 1:
            break
        reso //= 2
    assert [r for r, _ in levels] == [8, 4, 2, 1] and levels[0][1] == 64 ** 3       # 0, 8, .., 504 on each axis: all below the never-evaluated last plane
    sdf, n_eval = mesh_util.eval_grid_device(netMR, calib_t, R, chunk, True)
    assert n_eval == sum(n for _, n in levels)                      # the fused loop evaluated the same points ...
    assert torch.equal(sdf, sdf2)                                   # ... and left the identical field
    del sdf2, notproc
    observed('config3_512_octree', points_evaluated=n_eval, share_of_grid=n_eval / R ** 3, n_verts=len(verts),
             total_s_first_call=t1['total_s'], total_s=t2['total_s'], **{'max_err_vs_oracle_' + k: v for k, v in worst.items()})

    # (5) the mesh gen_mesh returned is the marching cubes of this field
    gv, gf = mesh_util.marching_cubes_device(ctx, sdf, 0.5)
    assert len(gv) == len(verts) and np.array_equal(gf.cpu().numpy()[:, ::-1], faces)
    # (4) a 128^3 crop around the centre of the inside region against the serial oracle
    inside = (sdf > 0.5).nonzero()
    ctr = [int(inside[:, k].float().mean().item()) for k in range(3)]
    lo = [min(max(c - 64, 0), R - 128) for c in ctr]
    crop = sdf[lo[0]:lo[0] + 128, lo[1]:lo[1] + 128, lo[2]:lo[2] + 128].contiguous()
    cv, cf = mesh_util.marching_cubes_device(ctx, crop, 0.5)
    ov, of_ = O.marching_cubes_lewiner(crop.cpu().numpy(), 0.5)
    assert len(ov) > 10000
    assert np.array_equal(cf.cpu().numpy(), of_) and np.array_equal(cv.cpu().numpy().view(np.uint32), ov.view(np.uint32))
