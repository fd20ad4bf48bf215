"""The configuration of the published pifuhd.pt - an MLP without norm layers (`MLP.py:37-41,65-69`) - on the lattice paths:
the library regroups the work (whole 128-row tiles, whole lattice columns, no statistics passes: `pifu_effective_chunk`)
instead of cutting it into the reference's 5 000-point query() calls.  Pinned here: (a) the regrouped evaluation is the
reference's field (CPU oracle in 5 000-point calls, 1e-4); (b) a split of the lattice over ranks at the library's granularity -
raw Z-slabs of the voxel grid, no chunk padding - reproduces the single-GPU field BIT FOR BIT, for dense ranges and for octree
index lists; (c) with a norm that couples a chunk nothing is regrouped."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import calib_tensor, golden_nets, seeded_features

pytestmark = pytest.mark.gpu
_SD_CACHE = {}      # seeded state dicts of this module's builds (seconds each; the tensors are never written in place)


def _build(mlp_norm, mlp_std=None):
    from pifuhd_amd import nets, synth, config
    sd, oc, of = golden_nets(mlp_norm=mlp_norm)
    if mlp_std is not None:      # a field with contrast for the un-normalised MLP (make_golden_r5.py: 0.10)
        oc = config.coarse_opt(mlp_norm=mlp_norm); of = config.fine_opt(oc)
        if (mlp_norm, mlp_std) not in _SD_CACHE:
            _SD_CACHE[(mlp_norm, mlp_std)] = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=1, mlp_std=mlp_std, perturb=True)
        sd = dict(_SD_CACHE[(mlp_norm, mlp_std)])
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(7)
    netG.set_features(fc)
    netMR.set_features(ff)
    return netG, netMR, sd, oc, of


def _eval(netMR, res, first, count, chunk, flags=0, index=None):
    from pifuhd_amd import _lib, mesh_util
    from pifuhd_amd.nets import _ptr, _stream_ptr
    calib, calib_inv = mesh_util.calib_pair(calib_tensor())
    out = torch.empty(count, device='cuda')
    res_a = (C.c_int32 * 3)(*res)
    _lib.call('pifu_eval_lattice', netMR._ctx.h, res_a, calib_inv.ctypes.data_as(C.c_void_p), calib.ctypes.data_as(C.c_void_p),
              _ptr(index) if index is not None else None, first, count, chunk, flags, _ptr(out), _stream_ptr(netMR.device))
    return out


def test_effective_chunk_follows_the_norms():
    from pifuhd_amd import _lib, dist as pdist
    _, net_no, *_ = _build('no')
    _, net_gn, *_ = _build('group')
    _, net_bn, *_ = _build('batch')          # netG eval (running statistics: pointwise), netMR train (batch statistics: coupled)
    assert pdist.effective_chunk(net_gn._ctx, 5000, 256) == 5000 and pdist.effective_chunk(net_gn._ctx, 5000, 0) == 5000
    assert pdist.effective_chunk(net_no._ctx, 5000, 0) == _lib.POINTWISE_CHUNK == 16384
    assert pdist.effective_chunk(net_no._ctx, 5000, 256) == 16384 and pdist.effective_chunk(net_no._ctx, 100000, 512) == 16384
    assert pdist.effective_chunk(net_no._ctx, 5000, 48) == 43 * 384          # lcm(128, 48) = 384: whole tiles AND whole columns
    assert pdist.effective_chunk(net_no._ctx, 5000, 100) == 5 * 3200
    assert pdist.effective_chunk(net_no._ctx, 5000, 500) == 16000 and pdist.effective_chunk(net_no._ctx, 5000, 513) == 16384 and pdist.effective_chunk(net_no._ctx, 5000, 1021) == 16384          # lcm 130 688: tiles only
    assert pdist.effective_chunk(net_bn._ctx, 5000, 256) == 5000
    assert pdist.effective_chunk(net_bn._ctx, 5000, 256, _lib.Q_COARSE_ONLY) == 16384


@pytest.mark.parametrize('flags_name', ['default', 'no_depth_walk', 'canonical'])
def test_pointwise_regrouping_is_the_reference_field(observed, flags_name):
    """61 440 lattice points in the un-normalised configuration: the library's own grouping against the CPU oracle evaluating
    the reference's 5 000-point query() calls."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib, mesh_util
    netG, netMR, sd, oc, of = _build('no', mlp_std=0.10)
    flags = {'default': 0, 'no_depth_walk': _lib.Q_NO_DEPTH_WALK, 'canonical': _lib.Q_NO_COLUMN_SHARING}[flags_name]
    res = (40, 32, 48)
    n = res[0] * res[1] * res[2]
    got = _eval(netMR, res, 0, n, 5000, flags | _lib.Q_FULL_COARSE).cpu().numpy()
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    ref.feat_coarse, ref.feat_fine = seeded_features(7)
    calib = calib_tensor()
    _, calib_inv = mesh_util.calib_pair(calib)
    want = []
    with torch.no_grad():
        for s in range(0, n, 5000):
            flat = np.arange(s, min(s + 5000, n), dtype=np.int64)
            pts = torch.from_numpy(O.grid_points_rect(res, calib_inv, flat).astype(np.float32))[None]
            want.append(ref.query(pts, calib)[0, 0].numpy())
    want = np.concatenate(want)
    err = float(np.abs(got - want).max())
    observed(f'pointwise_regrouping[{flags_name}]', gpu_vs_oracle=err, field_std=float(want[want > 0].std()))
    assert want[want > 0].std() > 0.02          # a field with contrast, not a constant
    assert np.array_equal(got == 0, want == 0) and err < 1e-4


@pytest.mark.parametrize('res', [(5, 7, 9), (3, 4, 131), (2, 3, 257), (1, 1, 5), (16, 16, 1), (2, 9, 1021)])
def test_pointwise_regrouping_on_ragged_lattices(res):
    """Lattices whose columns do not divide anything: 9, 131 (lcm with the 128-row tile 16 768: one group), 257 nodes per column,
    1 021 (lcm 130 688: the groups cut the columns), a single column of 5, columns of ONE node; default path (column sharing + depth walk where it applies) and the per-point path
    against the CPU oracle, the last group ragged in every case."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib, mesh_util
    netG, netMR, sd, oc, of = _build('no', mlp_std=0.10)
    n = res[0] * res[1] * res[2]
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    ref.feat_coarse, ref.feat_fine = seeded_features(7)
    calib = calib_tensor()
    _, calib_inv = mesh_util.calib_pair(calib)
    pts = torch.from_numpy(O.grid_points_rect(res, calib_inv, np.arange(n, dtype=np.int64)).astype(np.float32))[None]
    with torch.no_grad():
        want = ref.query(pts, calib)[0, 0].numpy()
    for flags in (0, _lib.Q_NO_DEPTH_WALK, _lib.Q_NO_COLUMN_SHARING):
        got = _eval(netMR, res, 0, n, 5000, flags | _lib.Q_FULL_COARSE).cpu().numpy()
        assert np.array_equal(got == 0, want == 0) and float(np.abs(got - want).max()) < 1e-4, flags
    # a range that starts and ends inside columns
    if n > 7:
        first, count = 3, n - 5
        got = _eval(netMR, res, first, count, 5000, _lib.Q_FULL_COARSE).cpu().numpy()
        assert float(np.abs(got - want[first:first + count]).max()) < 1e-4


@pytest.mark.parametrize('precision,bound', [(1, 1e-4), (2, 3e-5)])
def test_split_precision_on_the_regrouped_lattice(precision, bound):
    """The opt-in split products (1: bf16 x 3, 2: fp16 x 3; `pifu_set_precision`) where the MLP carries no norm layers: lattice
    evaluation in the library's own groups - column tables, depth walk, no statistics, the Cout = 1 layers as launches of their
    own (the fused epilogues are exact-fp32 only) - and an index list, against the exact-fp32 run of the same call."""
    from pifuhd_amd import _lib
    netG, netMR, *_ = _build('no', mlp_std=0.10)
    res = (24, 32, 48)
    n = res[0] * res[1] * res[2]
    index = (torch.rand(n, generator=torch.Generator().manual_seed(3)) < 0.4).nonzero()[:, 0].to('cuda', torch.int64).contiguous()
    exact = _eval(netMR, res, 0, n, 5000, _lib.Q_FULL_COARSE)
    exact_l = _eval(netMR, res, 0, int(index.numel()), 5000, index=index)
    _lib.call('pifu_set_precision', netMR._ctx.h, precision)
    try:
        got = _eval(netMR, res, 0, n, 5000, _lib.Q_FULL_COARSE)
        got_l = _eval(netMR, res, 0, int(index.numel()), 5000, index=index)
    finally:
        _lib.call('pifu_set_precision', netMR._ctx.h, 0)
    d, dl = float((got - exact).abs().max()), float((got_l - exact_l).abs().max())
    assert 0 < d < bound and 0 < dl < bound, (d, dl)
    assert torch.equal(_eval(netMR, res, 0, n, 5000, _lib.Q_FULL_COARSE), exact)          # and back to exact fp32


_SHAPES = {
    # a three-layer coarse MLP merging after its layer 1 - the 128-wide layer in front of the Cout = 1 layer, which is therefore
    # NOT taken into that layer's epilogue (its output is phi) - and a fine MLP whose last two hidden layers are 128 wide
    'short': dict(coarse=[257, 512, 128, 1], coarse_res=[1], merge=1, fine=[144, 256, 128, 128, 1], fine_res=[1, 2]),
    # the published widths with phi taken one layer earlier: phi is the 512-wide layer the depth walk produces
    # merge_layer -1 = the MLP's own default, len(dims) // 2 = layer 3 (MLP.py:25): phi is the 128-wide layer in front of the last one
    'merge_default': dict(coarse=[257, 1024, 512, 256, 128, 1], coarse_res=[2, 3, 4], merge=-1, fine=[144, 512, 256, 128, 1], fine_res=[1, 2]),
    'merge1': dict(coarse=[257, 1024, 512, 256, 128, 1], coarse_res=[2, 3, 4], merge=1, fine=[528, 512, 256, 128, 1], fine_res=[1, 2]),
}


@pytest.mark.parametrize('shape', sorted(_SHAPES))
@pytest.mark.parametrize('mlp_norm', ['no', 'group'])
def test_other_mlp_shapes_are_the_reference_field(observed, mlp_norm, shape):
    """MLP shapes other than the published ones (`options.py:96-108` are options, not constants), with and without norm layers,
    against the CPU oracle evaluating the reference's 5 000-point query() calls."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib, config, mesh_util, nets, synth
    sh = _SHAPES[shape]
    oc = config.coarse_opt(mlp_norm=mlp_norm, mlp_dim=sh['coarse'], mlp_res_layers=sh['coarse_res'], merge_layer=sh['merge'])
    of = config.fine_opt(oc, mlp_dim=sh['fine'], mlp_res_layers=sh['fine_res'])
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=11, mlp_std=0.10 if mlp_norm == 'no' else 0.16, perturb=True)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    fc, ff = seeded_features(7)
    netG.set_features(fc)
    netMR.set_features(ff)
    res = (24, 32, 48)
    n = res[0] * res[1] * res[2]
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    ref.feat_coarse, ref.feat_fine = fc, ff
    calib = calib_tensor()
    _, calib_inv = mesh_util.calib_pair(calib)
    want = []
    with torch.no_grad():
        for s in range(0, n, 5000):
            flat = np.arange(s, min(s + 5000, n), dtype=np.int64)
            pts = torch.from_numpy(O.grid_points_rect(res, calib_inv, flat).astype(np.float32))[None]
            want.append(ref.query(pts, calib)[0, 0].numpy())
    want = np.concatenate(want)
    for name, flags in (('default', 0), ('canonical', _lib.Q_NO_COLUMN_SHARING)):
        got = _eval(netMR, res, 0, n, 5000, flags | _lib.Q_FULL_COARSE).cpu().numpy()
        err = float(np.abs(got - want).max())
        observed(f'other_mlp_shapes[{shape},{mlp_norm},{name}]', gpu_vs_oracle=err, field_std=float(want[want > 0].std()))
        assert np.array_equal(got == 0, want == 0) and err < 1e-4
    assert want[want > 0].std() > 0.01


@pytest.mark.parametrize('ws', [2, 3, 8])
def test_pointwise_slab_split_is_bit_identical(ws):
    """What `dist.sharded_eval` does with N ranks, rank by rank in one process: every rank's share - whole groups of the
    library's own granularity, i.e. raw axis-0 slabs of the dense grid - evaluated by its own call, the shares side by side
    against the single call.  Dense range (columns of 48 nodes: groups of 43 x 384 points) and an octree-style index list."""
    from pifuhd_amd import dist as pdist
    netG, netMR, *_ = _build('no', mlp_std=0.10)
    res = (64, 32, 48)
    n = res[0] * res[1] * res[2]
    whole = _eval(netMR, res, 0, n, 5000)
    eff = pdist.effective_chunk(netMR._ctx, 5000, res[2])
    parts = []
    for r in range(ws):
        first, count, per = pdist.shard_chunks(n, eff, ws, r)
        assert first % res[2] == 0 and (count % eff == 0 or first + count == n)
        if count:
            parts.append(_eval(netMR, res, first, count, 5000))
    assert torch.equal(torch.cat(parts), whole)
    # index list: every third lattice node plus a dense stretch (long and short column segments for the depth walk)
    g = torch.Generator().manual_seed(5)
    keep = (torch.rand(n, generator=g) < 0.3)
    keep[20000:45000] = True
    index = keep.nonzero()[:, 0].to('cuda', torch.int64).contiguous()
    m = int(index.numel())
    whole_l = _eval(netMR, res, 0, m, 5000, index=index)
    eff_l = pdist.effective_chunk(netMR._ctx, 5000, 0)
    parts = []
    for r in range(ws):
        first, count, per = pdist.shard_chunks(m, eff_l, ws, r)
        if count:
            parts.append(_eval(netMR, res, first, count, 5000, index=index))
    assert torch.equal(torch.cat(parts), whole_l)
    # and the list evaluation is the dense one at those nodes up to fp32 summation order (segments are cut differently)
    assert float((whole_l - whole[index]).abs().max()) < 2e-5


def test_group_norm_is_never_regrouped():
    """With GroupNorm the reference's chunk is part of the numbers: a 5 000-point cut and a 4 096-point cut differ, and the
    library keeps the caller's."""
    netG, netMR, *_ = _build('group')
    res = (16, 32, 48)
    n = res[0] * res[1] * res[2]
    a = _eval(netMR, res, 0, n, 5000)
    b = _eval(netMR, res, 0, n, 4096)
    assert float((a - b).abs().max()) > 1e-6


def test_pointwise_octree_composition_vs_oracle(observed):
    """The multi-level octree of `reconstruction()` in the un-normalised configuration - every level's index list evaluated in
    the library's own groups, the Cout = 1 layers in the GEMM epilogues, the depth walk - against the same composition on the CPU
    oracle in 5 000-point calls (`mesh_util.py:124-187` + `PIFuMRNet.query`): R = 128, two levels, the same number of points per
    level, the float64 SDF within the north star's 1e-4, marching cubes on the device SDF bit-identical to the serial oracle's."""
    import os
    from oracle import pifu_oracle as O
    from pifuhd_amd import mesh_util, synth
    netG, netMR, sd, oc, of = _build('no', mlp_std=0.10)
    calib_t = calib_tensor()
    shift = synth.calibrate_occupancy_bias(netMR, calib_t, sd, 0.3, resolution=32)      # the 0.5 level inside the volume
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    ref.feat_coarse, ref.feat_fine = seeded_features(7)
    _, cinv = mesh_util.calib_pair(calib_t)

    def field(pts):
        with torch.no_grad():
            return ref.query(torch.from_numpy(pts.astype(np.float32))[None], calib_t)[0, 0].numpy()
    stats = []
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    try:
        want = O.eval_grid_octree(128, cinv, field, 5000, init_resolution=64, stats=stats)
    finally:
        torch.set_num_threads(threads)
    sdf, n_eval = mesh_util.eval_grid_device(netMR, calib_t, 128, 5000, True, init_resolution=64)
    got = sdf.cpu().numpy()
    err = float(np.abs(got - want).max())
    observed('pointwise_octree_128', gpu_vs_oracle_sdf=err, points=int(n_eval), bias_shift=shift)
    assert len(stats) == 2 and all(n > 0 for _, n in stats) and n_eval == sum(n for _, n in stats)
    assert err < 1e-4
    gv, gf = mesh_util.marching_cubes_device(netMR._ctx, sdf, 0.5)
    ov, of_ = O.marching_cubes_lewiner(got, 0.5)
    assert len(gv) > 1000 and np.array_equal(gf.cpu().numpy(), of_)
    assert np.array_equal(gv.cpu().numpy().view(np.uint32), ov.view(np.uint32))
