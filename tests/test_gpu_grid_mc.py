"""HIP grid evaluation (dense + octree) and marching cubes vs the oracle / golden fixtures.
Integer and index work is held to bit-exactness."""
import hashlib
import os

import numpy as np
import pytest
import torch

from helpers import calib_tensor, golden_nets, mlp_only_nets, noisy_field, seeded_features

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def mc_gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'mc_small.npz'))


@pytest.fixture(scope='module')
def ctx():
    from pifuhd_amd.nets import _Ctx
    return _Ctx('cuda')


def _mc(ctx, vol, level=0.5):
    from pifuhd_amd import mesh_util
    v, f = mesh_util.marching_cubes_device(ctx, torch.from_numpy(np.ascontiguousarray(vol)).cuda(), level)
    return v.cpu().numpy(), f.cpu().numpy()


def _same(v, f, gv, gf):
    return (v.shape == gv.shape and f.shape == gf.shape and np.array_equal(f, gf)
            and np.array_equal(v.view(np.uint32), gv.view(np.uint32)))


def test_mc_known_answers_noise_and_ellipsoid(ctx, mc_gold):
    g = mc_gold
    for i in range(int(g['kat_n'][0])):
        v, f = _mc(ctx, g[f'kat{i}_vol'])
        assert _same(v, f, g[f'kat{i}_v'], g[f'kat{i}_f']), f'kat{i}'
    for i in range(int(g['noise_n'][0])):
        v, f = _mc(ctx, g[f'noise{i}_vol'])
        assert _same(v, f, g[f'noise{i}_v'], g[f'noise{i}_f']), f'noise{i}'
    v, f = _mc(ctx, g['ell40_vol'])
    assert _same(v, f, g['ell40_v'], g['ell40_f'])
    # float64 input is cast to float32 first, as scikit-image does
    v, f = _mc(ctx, g['ell40_vol'].astype(np.float64))
    assert _same(v, f, g['ell40_v'], g['ell40_f'])


def test_mc_tie_cells(ctx, mc_gold):
    """4000 single cells with one-decimal values: exercises every eps / >= branch of the Lewiner tests."""
    g = mc_gold
    ov = np.concatenate([[0], np.cumsum(g['cells_nv'])])
    of = np.concatenate([[0], np.cumsum(g['cells_nf'])])
    bad = []
    for i, c in enumerate(g['cells_vol']):
        try:
            v, f = _mc(ctx, c)
        except (ValueError, RuntimeError):
            v, f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
        if not _same(v, f, g['cells_v'][ov[i]:ov[i + 1]], g['cells_f'][of[i]:of[i + 1]]):
            bad.append(i)
    assert not bad, bad[:10]


def test_mc_errors(ctx):
    with pytest.raises(ValueError):
        _mc(ctx, np.zeros((4, 4, 4), np.float32))
    with pytest.raises(RuntimeError):
        _mc(ctx, np.full((4, 4, 4), 0.5, np.float32))


def test_mc_reference_network_sdfs(ctx, mc_gold, golden_dir):
    r = np.load(os.path.join(golden_dir, 'recon_small.npz'))
    for tag in ('dense32', 'oct64'):
        v, f = _mc(ctx, r[f'{tag}_sdf_f32'])
        assert [len(v), len(f)] == list(mc_gold[f'{tag}_nv_nf'])
        assert hashlib.sha256(v.tobytes()).digest() == mc_gold[f'{tag}_v_sha'].tobytes()
        assert hashlib.sha256(f.tobytes()).digest() == mc_gold[f'{tag}_f_sha'].tobytes()


def test_mc_large_vs_oracle(ctx):
    """128^3 analytic body + noise: faces and vertices bit-exact against the serial oracle."""
    from oracle import pifu_oracle as O
    rs = np.random.RandomState(5)
    R = 128
    lin = -1 + 2 * np.arange(R) / R
    X, Y, Z = np.meshgrid(lin, lin, lin, indexing='ij')
    r = np.sqrt((0.8 * X) ** 2 + (0.45 * Y) ** 2 + (1.6 * Z) ** 2)
    vol = (1.0 / (1.0 + np.exp(-(0.35 - r) / 0.02)) + 0.05 * (rs.rand(R, R, R) - 0.5)).astype(np.float32)
    v, f = _mc(ctx, vol)
    vr, fr = O.marching_cubes_lewiner(vol, 0.5)
    assert _same(v, f, vr, fr)


def _mc_slabs(ctx, vol, nslabs, level=0.5):
    from pifuhd_amd import mesh_util
    v, f = mesh_util.marching_cubes_slabs(ctx, torch.from_numpy(np.ascontiguousarray(vol)).cuda(), level, nslabs)
    return v.cpu().numpy(), f.cpu().numpy()


def test_slab_marching_cubes_equals_the_whole_volume(ctx, mc_gold, golden_dir):
    """marching cubes over axis-0 slabs (vertex ownership and numbering of the whole volume, one halo plane of
    vertex numbers between neighbours) concatenates to the single-volume output bit for bit - on scikit-image's golden
    volumes, for slab counts that divide the cell layers unevenly, down to one cell layer per slab and to more slabs than
    cell layers."""
    g = mc_gold
    for i in range(int(g['noise_n'][0])):
        vol = g[f'noise{i}_vol']
        for ns in (1, 2, 3, vol.shape[0] - 1, vol.shape[0] + 3):
            v, f = _mc_slabs(ctx, vol, ns)
            assert _same(v, f, g[f'noise{i}_v'], g[f'noise{i}_f']), (i, ns)
    for ns in (2, 5, 39):
        v, f = _mc_slabs(ctx, g['ell40_vol'].astype(np.float64), ns)
        assert _same(v, f, g['ell40_v'], g['ell40_f']), ns
    r = np.load(os.path.join(golden_dir, 'recon_small.npz'))
    v, f = _mc_slabs(ctx, r['oct64_sdf_f32'], 8)
    assert hashlib.sha256(v.tobytes()).digest() == mc_gold['oct64_v_sha'].tobytes()
    assert hashlib.sha256(f.tobytes()).digest() == mc_gold['oct64_f_sha'].tobytes()
    # a non-cubic noisy body, 7 slabs, against the whole-volume kernel path
    rs = np.random.RandomState(11)
    lin = [-1 + 2 * np.arange(n) / n for n in (70, 44, 52)]
    X, Y, Z = np.meshgrid(*lin, indexing='ij')
    vol = (1.0 / (1.0 + np.exp(-(0.5 - np.sqrt(X ** 2 + (0.7 * Y) ** 2 + (1.3 * Z) ** 2)) / 0.03))
           + 0.2 * (rs.rand(70, 44, 52) - 0.5))
    want = _mc(ctx, vol)
    for ns in (2, 7, 8):
        assert _same(*_mc_slabs(ctx, vol, ns), *want), ns
    # the two exceptions come from the totals over the slabs
    with pytest.raises(ValueError):
        _mc_slabs(ctx, np.zeros((6, 4, 4), np.float32), 3)
    with pytest.raises(RuntimeError):
        _mc_slabs(ctx, np.full((6, 4, 4), 0.5, np.float32), 3)
    # a surface that lives in one slab only: the other slabs contribute nothing, their ranges still count
    vol = np.zeros((12, 6, 6), np.float32)
    vol[9, 3, 3] = 1.0
    assert _same(*_mc_slabs(ctx, vol, 4), *_mc(ctx, vol))


def test_slab_marching_cubes_on_a_1024_cube_volume(ctx):
    """BASELINE config 5's grid size on one GPU: a 1024^3 float64 SDF (8.6 GB, built on the device) meshed as 8 axis-0 slabs
    - what 8 ranks would each do with 1/8 of the index volume - equals the single-volume marching cubes bit for bit
    (vertex numbers up to ~2.5 M cross the slab boundaries through the halo planes)."""
    from pifuhd_amd import mesh_util
    R = 1024
    lin = torch.linspace(-1.0, 1.0 - 2.0 / R, R, device='cuda', dtype=torch.float64)
    x, y, z = lin[:, None, None], lin[None, :, None], lin[None, None, :]
    vol = torch.sigmoid((0.45 - torch.sqrt((0.9 * x) ** 2 + (0.5 * y) ** 2 + (1.4 * z) ** 2)) / 0.01)
    vol += 0.02 * torch.sin(91.0 * x + 40.0 * y) * torch.cos(67.0 * z)            # ripples: ambiguous cells near the surface
    del x, y, z
    v, f = mesh_util.marching_cubes_device(ctx, vol, 0.5)
    vs, fs = mesh_util.marching_cubes_slabs(ctx, vol, 0.5, 8)
    assert len(v) > 1_000_000 and torch.equal(f, fs) and torch.equal(v.view(torch.int32), vs.view(torch.int32))


def test_slab_marching_cubes_call_order_is_checked(ctx):
    from pifuhd_amd import _lib, mesh_util
    vol = torch.rand(8, 8, 8, device='cuda')
    step = mesh_util.SlabMarchingCubes(ctx, vol, 0.5, 2, 4)
    with pytest.raises(_lib.PifuError, match='pifu_mc_slab_count first'):
        step.verts(0)
    step.count()
    with pytest.raises(_lib.PifuError, match='pifu_mc_slab_verts first'):
        step.faces(None)
    _, halo = step.verts(0)
    with pytest.raises(_lib.PifuError, match='needs a halo'):
        step.faces(None)
    step.faces(halo)


@pytest.mark.parametrize('R,init', [(32, 4), (64, 8), (64, 16), (64, 64), (48, 6)])
def test_octree_bookkeeping_bit_exact(golden_dir, R, init):
    """The device octree (select / commit kernels) with the analytic field of the golden generator:
    the float64 SDF must be bit-identical to the reference's sequential numpy implementation."""
    from pifuhd_amd import mesh_util
    gold = np.load(os.path.join(golden_dir, 'octree_small.npz'))
    coords, _ = mesh_util.create_grid(R, R, R)
    calib_inv = np.linalg.inv(np.diag([1.0, -1.0, 1.0, 1.0]))
    c = coords.reshape(3, -1).T
    c = np.matmul(np.concatenate([c, np.ones((c.shape[0], 1))], 1), calib_inv.T)[:, :3].T.reshape(3, R, R, R)
    counts = []

    def f(p):
        counts.append(p.shape[1])
        return noisy_field(p)
    sdf = mesh_util.eval_grid_octree(c, f, init_resolution=init, threshold=0.05, num_samples=5000)
    assert hashlib.sha256(sdf.tobytes()).digest() == gold[f'oct_{R}_{init}_sha'].tobytes()
    assert np.array_equal(np.array(counts), gold[f'oct_{R}_{init}_chunks'])


@pytest.mark.parametrize('R,init', [(96, 16), (48, 16), (80, 16), (56, 8), (50, 7), (72, 24)])
def test_octree_bookkeeping_bit_exact_at_other_ratios(golden_dir, R, init):
    """resolution / init_resolution not a power of two (`--resolution 384 / 320 / 448 / 500` with the fixed init_resolution 64:
    the step halves through odd values, 50 // 7 leaves a remainder): the float64 SDF and the per-call point counts of the
    reference (`octree_ratios.npz`, generated by importing it: tests/golden/make_golden_r6.py)."""
    from pifuhd_amd import mesh_util
    gold = np.load(os.path.join(golden_dir, 'octree_ratios.npz'))
    coords, _ = mesh_util.create_grid(R, R, R)
    calib_inv = np.linalg.inv(np.diag([1.0, -1.0, 1.0, 1.0]))
    c = coords.reshape(3, -1).T
    c = np.matmul(np.concatenate([c, np.ones((c.shape[0], 1))], 1), calib_inv.T)[:, :3].T.reshape(3, R, R, R)
    counts = []

    def f(p):
        counts.append(p.shape[1])
        return noisy_field(p)
    sdf = mesh_util.eval_grid_octree(c, f, init_resolution=init, threshold=0.05, num_samples=5000)
    assert hashlib.sha256(sdf.tobytes()).digest() == gold[f'oct_{R}_{init}_sha'].tobytes()
    assert np.array_equal(np.array(counts), gold[f'oct_{R}_{init}_chunks'])


def test_network_grid_vs_reference_sdf(golden_dir):
    """Dense 32^3 and octree 64^3 with the network field: SDF within 1e-4 of the reference's, and the
    triangle index stream bit-exact against the serial oracle run on the SAME device SDF."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import mesh_util, nets
    gold = np.load(os.path.join(golden_dir, 'recon_small.npz'))
    sd, oc, of = golden_nets()
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(9)
    netG.set_features(fc)
    netMR.set_features(ff)
    calib = calib_tensor()
    for tag, R, octree in (('dense32', 32, False), ('oct64', 64, True)):
        res = mesh_util.reconstruction(netMR, 'cuda', calib, R, None, None, 0.5, use_octree=octree, num_samples=5000,
                                       return_sdf=True)
        assert res != -1
        verts, faces, _, _, sdf = res
        sdf = sdf.cpu().numpy()
        assert np.abs(sdf - gold[f'{tag}_sdf_f32']).max() < 1e-4
        vr, fr = O.marching_cubes_lewiner(sdf, 0.5)
        assert np.array_equal(faces[:, ::-1], fr)
        # vertex back-transform (mesh_util.py:87-89) in float64
        calib_inv = np.linalg.inv(calib[0].numpy().astype(np.float64))
        mat = np.eye(4); mat[0, 0] = mat[1, 1] = mat[2, 2] = 2.0 / R; mat[:3, 3] = -1
        tm = calib_inv @ mat
        assert np.array_equal(verts, (tm[:3, :3] @ vr.T.astype(np.float64) + tm[:3, 3:4]).T)


def test_obj_writer(tmp_path):
    from oracle import pifu_oracle as O
    from pifuhd_amd import mesh_util
    rs = np.random.RandomState(3)
    v = rs.randn(500, 3) * 3
    c = rs.rand(500, 3)
    f = rs.randint(0, 500, size=(900, 3)).astype(np.int32)
    p = tmp_path / 'm.obj'
    mesh_util.save_obj_mesh_with_color(str(p), v, f, c)
    assert p.read_text() == O.obj_text(v, f, c)


def test_column_sharing_equals_per_point_evaluation():
    """Lattice evaluation shares the image-feature products of an (i, j) column; the result must agree with the
    independent per-point evaluation up to fp32 summation order, for dense ranges (incl. one that starts in the
    middle of a column) and for an octree-style index list."""
    import ctypes as C
    from pifuhd_amd import _lib, nets
    from pifuhd_amd.nets import _ptr, _stream_ptr
    sd, oc, of = golden_nets()
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(9)
    netG.set_features(fc)
    netMR.set_features(ff)
    calib = np.ascontiguousarray(calib_tensor()[0].numpy())
    cinv = np.ascontiguousarray(np.linalg.inv(calib.astype(np.float64)))
    res = (C.c_int32 * 3)(24, 40, 48)

    def run(flags, index, first, count):
        out = torch.empty(count, device='cuda')
        _lib.call('pifu_eval_lattice', netMR._ctx.h, res, cinv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p),
                  _ptr(index), first, count, 5000, flags, _ptr(out), _stream_ptr(netMR.device))
        return out.cpu().numpy()
    total = 24 * 40 * 48
    for first, count in ((0, total), (5000 * 3, 5000 * 2 + 777)):
        a = run(_lib.Q_FULL_COARSE, None, first, count)
        b = run(_lib.Q_FULL_COARSE | _lib.Q_NO_COLUMN_SHARING, None, first, count)
        assert np.abs(a - b).max() < 2e-5
        assert np.abs(a - b).max() > 0          # the two paths really are different computations
    g = torch.Generator().manual_seed(3)
    idx = torch.sort(torch.randperm(total, generator=g)[:12345])[0].cuda()
    a = run(0, idx, 0, 12345)
    b = run(_lib.Q_NO_COLUMN_SHARING, idx, 0, 12345)
    assert np.abs(a - b).max() < 2e-5


@pytest.mark.parametrize('mlp_norm', ['group', 'batch', 'no'])
def test_depth_walk_equals_gemm_paths(mlp_norm):
    """Coarse layer 1 by the depth walk (walk.inc) vs the same layer as a GEMM (Q_NO_DEPTH_WALK) vs the canonical
    per-point evaluation: dense ranges whose chunks cut columns, an octree-like shell (index list with long column
    segments, which takes the device-count path), and the coarse-only net."""
    import ctypes as C
    from pifuhd_amd import _lib, nets
    from pifuhd_amd.nets import _ptr, _stream_ptr
    sd, oc, of = golden_nets(mlp_norm=mlp_norm)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(5)
    netG.set_features(fc)
    netMR.set_features(ff)
    calib = np.ascontiguousarray(calib_tensor()[0].numpy())
    cinv = np.ascontiguousarray(np.linalg.inv(calib.astype(np.float64)))
    R0, R1, R2 = 20, 24, 56
    res = (C.c_int32 * 3)(R0, R1, R2)
    tol = 2e-5 if mlp_norm == 'group' else 1e-4

    def run(flags, index, first, count, chunk=700):
        out = torch.empty(count, device='cuda')
        _lib.call('pifu_eval_lattice', netMR._ctx.h, res, cinv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p),
                  _ptr(index), first, count, chunk, flags, _ptr(out), _stream_ptr(netMR.device))
        return out.cpu().numpy()
    total = R0 * R1 * R2
    for first, count in ((0, total), (700 * 3 + 0, 700 * 5 + 333)):
        w = run(_lib.Q_FULL_COARSE, None, first, count)
        g = run(_lib.Q_FULL_COARSE | _lib.Q_NO_DEPTH_WALK, None, first, count)
        p = run(_lib.Q_FULL_COARSE | _lib.Q_NO_COLUMN_SHARING, None, first, count)
        assert np.abs(w - g).max() < tol and np.abs(w - p).max() < tol
        assert np.abs(w - g).max() > 0                       # really a different computation
        assert np.array_equal(w, run(_lib.Q_FULL_COARSE, None, first, count))      # reproducible
    # shell: every column, k in [9, 41) -> 32-point segments; chunk 5000 cuts them
    k = torch.arange(9, 41)
    idx = (torch.arange(R0 * R1)[:, None] * R2 + k[None, :]).reshape(-1).cuda()
    w = run(0, idx, 0, idx.numel(), 5000)
    g = run(_lib.Q_NO_DEPTH_WALK, idx, 0, idx.numel(), 5000)
    p = run(_lib.Q_NO_COLUMN_SHARING, idx, 0, idx.numel(), 5000)
    assert np.abs(w - g).max() < tol and np.abs(w - p).max() < tol and np.abs(w - g).max() > 0
    # scattered points (segments of length ~1): the walk declines, the result is the GEMM path's bit for bit
    gsel = torch.Generator().manual_seed(5)
    sidx = torch.sort(torch.randperm(total, generator=gsel)[:3000])[0].cuda()
    assert np.array_equal(run(0, sidx, 0, 3000, 5000), run(_lib.Q_NO_DEPTH_WALK, sidx, 0, 3000, 5000))
    # coarse-only net
    w = run(_lib.Q_COARSE_ONLY, None, 0, total)
    p = run(_lib.Q_COARSE_ONLY | _lib.Q_NO_COLUMN_SHARING, None, 0, total)
    assert np.abs(w - p).max() < tol


@pytest.mark.parametrize('dims', [[257, 1024, 256, 256, 128, 1], [257, 512, 1024, 256, 128, 1], [257, 256, 512, 256, 128, 1]])
def test_depth_walk_with_other_layer_widths(dims):
    """The depth walk for coarse MLPs whose layer 1 is 256 wide (one output per thread), 1024 wide (two slices of 512) and
    whose layer 0 is 512 / 256 wide: same field as the GEMM form of layer 1 and as the per-point evaluation."""
    import ctypes as C
    from pifuhd_amd import _lib, config, nets, synth
    from pifuhd_amd.nets import _ptr, _stream_ptr
    oc = config.coarse_opt(mlp_dim=dims)
    of = config.fine_opt(oc)
    spec = {}
    synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
    synth.mlp_keys('mlp', of.mlp_dim, of.mlp_res_layers, of.mlp_norm, spec)
    sd = synth.name_seeded_state_dict(spec, seed=4, mlp_std=0.16, perturb=True)
    netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda')
    netMR = nets.PIFuMRNet(of, netG)
    netG.eval()
    netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}
    netG._push_weights()
    netMR._sd = {k: t.cuda() for k, t in sd.items() if not k.startswith('netG.')}
    netMR._push_weights()
    fc, ff = seeded_features(6)
    netG.set_features(fc)
    netMR.set_features(ff)
    calib = np.ascontiguousarray(calib_tensor()[0].numpy())
    cinv = np.ascontiguousarray(np.linalg.inv(calib.astype(np.float64)))
    R0, R1, R2 = 12, 16, 64
    res = (C.c_int32 * 3)(R0, R1, R2)
    total = R0 * R1 * R2

    def run(flags):
        out = torch.empty(total, device='cuda')
        _lib.call('pifu_eval_lattice', netMR._ctx.h, res, cinv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p),
                  None, 0, total, 900, flags, _ptr(out), _stream_ptr(netMR.device))
        return out.cpu().numpy()
    w, g, p = run(_lib.Q_FULL_COARSE), run(_lib.Q_FULL_COARSE | _lib.Q_NO_DEPTH_WALK), run(_lib.Q_FULL_COARSE | _lib.Q_NO_COLUMN_SHARING)
    assert np.abs(w - g).max() < 2e-5 and np.abs(w - p).max() < 2e-5
    assert np.abs(w - g).max() > 0                           # the walk really ran
    assert 0.02 < w.mean() < 0.98


def test_full_size_lattice_properties():
    """BASELINE configs[1] at full size (dense 256^3, chunk 5000) through size-independent properties: the field does
    not depend on how many rows go into one batch of launches, a chunk-aligned slab evaluated on its own (the unit the
    multi-GPU path shards by) is the corresponding slice of the full field bit for bit, reruns are bit-identical, and
    chunks sampled from the field (first, middle, ragged last) agree with the CPU oracle within the north star's 1e-4."""
    import ctypes as C
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib, nets
    from pifuhd_amd.nets import _ptr, _stream_ptr
    sd, oc, of = mlp_only_nets()
    netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda')
    netMR = nets.PIFuMRNet(of, netG)
    netG.eval()
    netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}
    netG._push_weights()
    netMR._sd = {k: t.cuda() for k, t in sd.items() if not k.startswith('netG.')}
    netMR._push_weights()
    fc, ff = seeded_features(21, hc=128, hf=512)
    netG.set_features(fc)
    netMR.set_features(ff)
    calib_t = calib_tensor()
    calib = np.ascontiguousarray(calib_t[0].numpy())
    cinv = np.ascontiguousarray(np.linalg.inv(calib.astype(np.float64)))
    R, chunk = 256, 5000
    res = (C.c_int32 * 3)(R, R, R)
    total = R ** 3
    h = netMR._ctx.h

    def run(first, count, rows):
        _lib.call('pifu_reserve', h, rows)
        out = torch.empty(count, device='cuda')
        _lib.call('pifu_eval_lattice', h, res, cinv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p),
                  None, first, count, chunk, _lib.Q_FULL_COARSE, _ptr(out), _stream_ptr(netMR.device))
        return out
    full = run(0, total, 2097152)
    assert torch.equal(full, run(0, total, 2097152))                      # reproducible
    assert torch.equal(full, run(0, total, 262144))                       # independent of the batch size
    first, count = 1000 * chunk, 300 * chunk                              # a slab of whole chunks that cuts columns
    assert torch.equal(full[first:first + count], run(first, count, 1048576))
    assert float(full.min()) >= 0.0 and float(full.max()) <= 1.0 and float(full.std()) > 1e-3
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    ref.feat_coarse, ref.feat_fine = fc, ff
    n_chunks = (total + chunk - 1) // chunk
    with torch.no_grad():
        for c in (0, n_chunks // 2, n_chunks - 1):
            lo, hi = c * chunk, min((c + 1) * chunk, total)
            pts = torch.from_numpy(O.grid_points_rect((R, R, R), cinv, np.arange(lo, hi, dtype=np.int64)).astype(np.float32))[None]
            want = ref.query(pts, calib_t)[0, 0].numpy()
            assert np.abs(full[lo:hi].cpu().numpy() - want).max() < 1e-4


def test_clock_probe_reports_a_plausible_shader_clock():
    """pifu_clock_probe (bench.py's measured-clock figure): two probes around continuous work give, per XCD sampled by
    both, shader ticks per 100 MHz real-time tick.  The shader counter of an XCD restarts when the XCD is power-gated, so
    the probes bracket a busy period (as in bench.py) and the median over the XCDs is what counts."""
    import ctypes as C
    from pifuhd_amd import _lib, nets
    from pifuhd_amd.nets import _stream_ptr
    sd, oc, of = mlp_only_nets()
    netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda')
    h = netG._ctx.h
    a, b = (C.c_uint64 * 16)(), (C.c_uint64 * 16)()
    x = torch.randn(8192, 8192, device='cuda')

    def burn(n):
        y = x
        for _ in range(n):
            y = torch.tanh(y @ x * 1e-4)
        return y
    burn(30)                                          # the GPU is busy and clocked up when the first probe runs
    _lib.call('pifu_clock_probe', h, a, _stream_ptr(netG.device))
    burn(60)
    _lib.call('pifu_clock_probe', h, b, _stream_ptr(netG.device))
    mhz = [100.0 * (b[2 * i] - a[2 * i]) / (b[2 * i + 1] - a[2 * i + 1]) for i in range(8) if a[2 * i + 1] and b[2 * i + 1] > a[2 * i + 1]]
    good = sorted(m for m in mhz if 100.0 < m < 4000.0)
    assert good, mhz
    assert 500.0 < good[len(good) // 2] < 3000.0, mhz


def test_lattice_slabs_at_the_far_end_of_a_1024_cube_grid():
    """BASELINE configs[4] index range: flat lattice indices close to 2^30.  Two slabs of a 1024^3 lattice (one that ends
    at the last point with a ragged final chunk, one in the middle that starts inside a column) against the CPU oracle
    evaluated on the same flat indices."""
    import ctypes as C
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib, mesh_util, nets
    from pifuhd_amd.nets import _ptr, _stream_ptr
    sd, oc, of = mlp_only_nets()
    netG = nets.PIFuNetwNML(oc, 'orthogonal', 'cuda')
    netMR = nets.PIFuMRNet(of, netG)
    netG.eval()
    netG._sd = {k[5:]: t.cuda() for k, t in sd.items() if k.startswith('netG.')}
    netG._push_weights()
    netMR._sd = {k: t.cuda() for k, t in sd.items() if not k.startswith('netG.')}
    netMR._push_weights()
    fc, ff = seeded_features(23, hc=128, hf=512)
    netG.set_features(fc)
    netMR.set_features(ff)
    calib_t = calib_tensor()
    calib, cinv = mesh_util.calib_pair(calib_t)
    R, chunk = 1024, 5000
    res = (C.c_int32 * 3)(R, R, R)
    total = R ** 3
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    ref.feat_coarse, ref.feat_fine = fc, ff
    for first, count in ((total - 2 * chunk - 1234, 2 * chunk + 1234), (total // 2 + 777, 3 * chunk)):
        out = torch.empty(count, device='cuda')
        _lib.call('pifu_eval_lattice', netMR._ctx.h, res, cinv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p),
                  None, first, count, chunk, _lib.Q_FULL_COARSE, _ptr(out), _stream_ptr(netMR.device))
        got = out.cpu().numpy()
        with torch.no_grad():
            for lo in range(0, count, chunk):
                hi = min(lo + chunk, count)
                flat = np.arange(first + lo, first + hi, dtype=np.int64)
                pts = torch.from_numpy(O.grid_points_rect((R, R, R), cinv, flat).astype(np.float32))[None]
                want = ref.query(pts, calib_t)[0, 0].numpy()
                assert np.abs(got[lo:hi] - want).max() < 1e-4, (first, lo)


def test_config3_octree_512_and_marching_cubes_match_the_reference(ctx, golden_dir):
    """BASELINE configs[2]: a 512^3 grid with octree refinement.  The analytic noisy field goes through the device octree
    (select / commit kernels, init_resolution 64, threshold 0.05, 5 000-point chunks); the float64 SDF must be
    sha256-identical to the one the REFERENCE's eval_grid_octree produced (tests/golden/make_golden_r2.py), with the same
    number of eval_func calls and points; marching cubes on it must reproduce scikit-image 0.18.3's vertex and face
    streams (tests/golden/make_golden_mc512.py) bit for bit, and so must the serial oracle."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import mesh_util
    gold = np.load(os.path.join(golden_dir, 'octree512.npz'))
    mcg = np.load(os.path.join(golden_dir, 'mc_oct512.npz'))
    R = 512
    calib_inv = np.linalg.inv(np.diag([1.0, -1.0, 1.0, 1.0]))
    counts = []

    def field(p):
        counts.append(p.shape[1])
        return noisy_field(p)
    sdf = mesh_util.eval_grid_octree(None, field, init_resolution=64, threshold=0.05, num_samples=5000, resolution=R,
                                     calib_inv=calib_inv)
    assert [len(counts), int(np.sum(counts))] == list(gold['chunks_total'])
    assert hashlib.sha256(np.array(counts, dtype=np.int64).tobytes()).digest() == gold['chunks_sha'].tobytes()
    assert np.array_equal(sdf[::16, ::16, ::16], gold['sub'])
    assert hashlib.sha256(sdf.tobytes()).digest() == gold['sha'].tobytes()
    v, f = _mc(ctx, sdf)
    assert [len(v), len(f)] == list(mcg['nv_nf'])
    assert np.array_equal(f[::997], mcg['f_sub']) and np.array_equal(v[::997].view(np.uint32), mcg['v_sub'].view(np.uint32))
    assert hashlib.sha256(f.tobytes()).digest() == mcg['f_sha'].tobytes()
    assert hashlib.sha256(v.tobytes()).digest() == mcg['v_sha'].tobytes()
    vr, fr = O.marching_cubes_lewiner(sdf, 0.5)            # the checker agrees with scikit-image at this size too
    assert _same(v, f, vr, fr)


@pytest.mark.parametrize('tag', ['scale_shift', 'rot_y'])
def test_dense_grid_with_nontrivial_calibrations(golden_dir, tag):
    """reconstruction() with a zoom + translation calib (lattice columns still share their image coordinates: the
    column-sharing and depth-walk path) and with a rotation about y (they do not: per-point path): the SDF must match the
    reference's, marching cubes on it the oracle's, and the vertex back-transform the reference's float64 formula with
    the float32 inverse."""
    from oracle import pifu_oracle as O
    from helpers import calib_variant
    from pifuhd_amd import mesh_util, nets
    gold = np.load(os.path.join(golden_dir, 'calib_variants.npz'))
    sd, oc, of = golden_nets()
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(9)
    netG.set_features(fc)
    netMR.set_features(ff)
    calib = calib_variant(tag)
    res = mesh_util.reconstruction(netMR, 'cuda', calib, 32, None, None, 0.5, use_octree=False, num_samples=5000,
                                   return_sdf=True)
    assert res != -1
    verts, faces, _, _, sdf = res
    sdf = sdf.cpu().numpy()
    assert np.abs(sdf - gold[f'{tag}_dense32_sdf_f32']).max() < 1e-4
    vr, fr = O.marching_cubes_lewiner(sdf, 0.5)
    c32 = calib[0].numpy()
    cinv = np.linalg.inv(c32)                               # float32, as the reference inverts it (mesh_util.py:61-62)
    mat = np.eye(4)
    mat[0, 0] = mat[1, 1] = mat[2, 2] = 2.0 / 32
    mat[0:3, 3] = -1.0
    tm = np.matmul(cinv, mat)
    want_v = (np.matmul(tm[:3, :3], vr.T.astype(np.float32)) + tm[:3, 3:4]).T
    assert np.array_equal(verts, want_v)
    assert np.array_equal(faces, fr[:, ::-1] if np.linalg.det(tm[:3, :3]) < 0 else fr)


def test_full_size_marching_cubes_properties(ctx):
    """Size-independent properties at BASELINE's grid size (512^3), where the serial oracle is too slow to be the checker:
    the mesh of a closed analytic body is ONE watertight component (every edge shared by exactly two faces:
    `pifu_face_components`), its Euler characteristic is 2 (V - F / 2 = 2 for a closed triangle mesh of genus 0), every face
    index is a valid vertex, the vertices lie within half a cell of the analytic surface, and a second run reproduces the
    arrays bit for bit."""
    from pifuhd_amd import _lib, mesh_util
    from pifuhd_amd.nets import _ptr, _stream_ptr
    R = 512
    lin = torch.linspace(-1.0, 1.0 - 2.0 / R, R, device='cuda', dtype=torch.float64)
    x, y, z = lin[:, None, None], lin[None, :, None], lin[None, None, :]
    rad = torch.sqrt((0.9 * x) ** 2 + (0.55 * y) ** 2 + (1.5 * z) ** 2)
    vol = torch.sigmoid((0.45 - rad) / 0.01)
    del x, y, z, rad
    v, f = mesh_util.marching_cubes_device(ctx, vol, 0.5)
    V, F = len(v), len(f)
    assert V > 200_000 and F % 2 == 0 and V - F // 2 == 2
    assert int(f.min()) == 0 and int(f.max()) == V - 1
    label = torch.empty(F, device='cuda', dtype=torch.int32)
    open_edge = torch.empty(F, device='cuda', dtype=torch.uint8)
    _lib.call('pifu_face_components', ctx.h, _ptr(f), F, _ptr(label), _ptr(open_edge), _stream_ptr(ctx.device))
    assert int(open_edge.sum()) == 0 and int(label.max()) == 0
    p = v.double() * (2.0 / R) - 1.0                        # array-axis order = (x, y, z) of the lattice above
    r = torch.sqrt((0.9 * p[:, 0]) ** 2 + (0.55 * p[:, 1]) ** 2 + (1.5 * p[:, 2]) ** 2)
    assert float((r - 0.45).abs().max()) < 1.5 * 1.5 * (2.0 / R)        # within a cell of the level set (largest axis scale 1.5)
    v2, f2 = mesh_util.marching_cubes_device(ctx, vol, 0.5)
    assert torch.equal(f, f2) and torch.equal(v.view(torch.int32), v2.view(torch.int32))


@pytest.mark.parametrize('R,init,n_levels', [(128, 64, 2), (128, 16, 4), (96, 16, 3)])
def test_multi_level_network_octree_composition_vs_oracle(R, init, n_levels):
    """The COMPOSITION `pifu_eval_grid` runs for gen_mesh - several octree levels, each level's points chosen from the SDF
    the previous levels left, evaluated by the network (column sharing, depth walk, 5000-point GroupNorm chunks cut from
    the level's own index list), committed and block-filled - against the same composition on the CPU oracle
    (`mesh_util.py:124-187` + `PIFuMRNet.query`): R = 128 with init_resolution 64 (two levels) and 16 (four levels), R = 96 with
    init_resolution 16 (steps 6, 3, 1: what `--resolution 384` does with the fixed init_resolution 64).  The
    levels must select the same number of points, the float64 SDF must agree within the north star's 1e-4, the fused
    device loop must equal the level-by-level ABI calls bit for bit, and marching cubes on the device SDF must be the
    serial oracle's on that SDF, vertices and faces."""
    import ctypes as C
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib, mesh_util, nets
    from pifuhd_amd.nets import _ptr, _stream_ptr
    chunk = 5000
    sd, oc, of = golden_nets()
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(9)
    netG.set_features(fc)
    netMR.set_features(ff)
    calib_t = calib_tensor()
    calib, cinv = mesh_util.calib_pair(calib_t)

    ref = O.OracleNets(sd, oc, of, fine_training=True)
    ref.feat_coarse, ref.feat_fine = fc, ff

    def field(pts):
        with torch.no_grad():
            return ref.query(torch.from_numpy(pts.astype(np.float32))[None], calib_t)[0, 0].numpy()
    stats = []
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))      # torch-CPU on these small GEMMs is slower with every host thread
    try:
        want = O.eval_grid_octree(R, cinv, field, chunk, init_resolution=init, stats=stats)
    finally:
        torch.set_num_threads(threads)
    assert len(stats) == n_levels and all(n > 0 for _, n in stats)

    # the fused device loop
    sdf, n_eval = mesh_util.eval_grid_device(netMR, calib_t, R, chunk, True, init_resolution=init)
    assert n_eval == sum(n for _, n in stats)
    got = sdf.cpu().numpy()
    assert np.abs(got - want).max() < 1e-4

    # the same composition level by level through the public octree entry points: per-level counts, identical SDF
    ctx = netMR._ctx
    st = _stream_ptr(ctx.device)
    res_a = (C.c_int32 * 3)(R, R, R)
    sdf2 = torch.empty(R, R, R, device='cuda', dtype=torch.float64)
    notproc = torch.empty(R, R, R, device='cuda', dtype=torch.uint8)
    _lib.call('pifu_octree_begin', ctx.h, R, _ptr(sdf2), _ptr(notproc), st)
    reso, levels = R // init, []
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
        _lib.call('pifu_octree_commit', ctx.h, R, reso, 0.05, _ptr(idx), _ptr(vals), n, _ptr(sdf2), _ptr(notproc), st)
        if reso <= 1:
            break
        reso //= 2
    assert levels == stats
    assert torch.equal(sdf2, sdf)

    # marching cubes on the device SDF: the serial oracle on the same volume, bit for bit
    gv, gf = mesh_util.marching_cubes_device(ctx, sdf, 0.5)
    ov, of_ = O.marching_cubes_lewiner(got, 0.5)
    assert len(ov) > 1000
    assert np.array_equal(gf.cpu().numpy(), of_) and np.array_equal(gv.cpu().numpy().view(np.uint32), ov.view(np.uint32))
