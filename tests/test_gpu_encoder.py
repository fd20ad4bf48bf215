"""HIP encoders (HGFilter + pix2pixHD normal nets as implicit-GEMM MFMA convolutions) vs the golden vectors
generated from the reference (tests/golden/filter_small.npz, filter_full_sub.npz)."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_nets

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))


def test_filter_tiny_matches_reference(golden_dir):
    """64x64 coarse image / 128x128 fine image, perturbed parameters; the fine level runs its BatchNorm with
    batch statistics because recon() leaves netMR in train mode (SURVEY C-2)."""
    from pifuhd_amd import nets, synth
    gold = np.load(os.path.join(golden_dir, 'filter_small.npz'))
    sd, oc, of = golden_nets()
    of.loadSizeBig = 128
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    assert netMR.training and not netG.training
    netMR.filter_global(synth.synthetic_image(64, 3, seed=2))
    assert _rel(netG.nmlF.cpu().numpy(), gold['tiny_nmlF']) < 1e-4
    assert _rel(netG.nmlB.cpu().numpy(), gold['tiny_nmlB']) < 1e-4
    fc = netG.im_feat_list[0].cpu().numpy()
    assert fc.shape == gold['tiny_feat_coarse'].shape
    assert _rel(fc, gold['tiny_feat_coarse']) < 2e-4
    netMR.filter_local(synth.synthetic_image(128, 3, seed=3)[:, None])
    ff = netMR.im_feat_list[0].cpu().numpy()
    assert ff.shape == gold['tiny_feat_fine_trainmode'].shape
    assert _rel(ff, gold['tiny_feat_fine_trainmode']) < 2e-4
    # and the query path consumes the NHWC maps the encoder installed: compare with re-installing the NCHW export
    calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
    g = torch.Generator().manual_seed(5)
    pts = (torch.rand(1, 3, 2000, generator=g) * 2 - 1)
    netMR.query(pts, calib)
    a = netMR.get_preds().clone()
    netG.set_features(netG.im_feat_list[0]); netMR.set_features(netMR.im_feat_list[0])
    netMR.query(pts, calib)
    assert torch.equal(a, netMR.get_preds())


def test_filter_full_size_subsamples(golden_dir):
    """512^2 filter_global and 1024^2 filter_local against strided sub-samples of the reference's outputs."""
    from pifuhd_amd import config, nets, synth
    gold = np.load(os.path.join(golden_dir, 'filter_full_sub.npz'))
    oc = config.coarse_opt(); of = config.fine_opt(oc)
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = synth.synthetic_inputs(1024, 3, seed=0)
    netMR.filter_global(data['img_512'])
    assert _rel(netG.nmlF[0, :, ::16, ::16].cpu().numpy(), gold['nmlF_sub']) < 2e-4
    assert _rel(netG.nmlB[0, :, ::16, ::16].cpu().numpy(), gold['nmlB_sub']) < 2e-4
    assert _rel(netG.im_feat_list[0][0, ::8, ::8, ::8].cpu().numpy(), gold['feat_coarse_sub']) < 5e-4
    netMR.filter_local(data['img'][:, None])
    assert _rel(netMR.im_feat_list[0][0, :, ::16, ::16].cpu().numpy(), gold['feat_fine_sub']) < 5e-4


def test_group_norm_filter_variant(golden_dir):
    """HGFilter with norm='group' (what the original PIFuHD checkpoints use), 'no_down', 9 -> 16 channels."""
    import ctypes as C
    from pifuhd_amd import _lib, synth
    from pifuhd_amd.nets import _Ctx, _ptr, _stream_ptr
    gold = np.load(os.path.join(golden_dir, 'filter_small.npz'))
    spec = {}
    synth.filter_keys('image_filter', 1, 2, 9, 16, 'group', 'no_down', spec)
    sd = synth.name_seeded_state_dict(spec, seed=5, perturb=True)
    ctx = _Ctx('cuda')
    for k, t in sd.items():
        a = np.ascontiguousarray(t.numpy().astype(np.float32))
        shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
        _lib.call('pifu_enc_set_param', ctx.h, ('fine.' + k).encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 9, 64, 64, generator=g).cuda()
    out = torch.empty(1, 16, 32, 32, device='cuda')
    _lib.call('pifu_filter_fine', ctx.h, _ptr(x), 9, 64, 64, None, None, 64, 64, 1, 2, 16, 2, _ptr(out), _stream_ptr(ctx.device))
    assert _rel(out.cpu().numpy(), gold['group_filter_out']) < 2e-4


def test_rgbd_six_channel_input(golden_dir):
    """BASELINE configs[1] says RGB-D: 6 image channels (RGB || depth, `train.py:173-174`) through
    `Filter(in_channels=6+6)` and `define_G(6, ...)` (SURVEY C-7), against goldens from the reference's parametric
    constructors; then a query on the resulting maps."""
    from helpers import calib_tensor, seeded_points
    from pifuhd_amd import config, nets, synth
    gold = np.load(os.path.join(golden_dir, 'rgbd_small.npz'))
    oc = config.coarse_opt(in_image_channels=6)
    of = config.fine_opt(oc)
    of.loadSizeBig = 128
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=1, mlp_std=0.16, perturb=True)
    assert sd['netG.image_filter.conv1.weight'].shape[1] == 12 and sd['netG.netF.model.1.weight'].shape[1] == 6
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    netMR.filter_global(synth.synthetic_image(64, 6, seed=4))
    assert _rel(netG.nmlF.cpu().numpy(), gold['nmlF']) < 1e-4
    assert _rel(netG.nmlB.cpu().numpy(), gold['nmlB']) < 1e-4
    assert _rel(netG.im_feat_list[0].cpu().numpy(), gold['feat_coarse']) < 2e-4
    netMR.filter_local(synth.synthetic_image(128, 6, seed=5)[:, None])
    assert _rel(netMR.im_feat_list[0].cpu().numpy(), gold['feat_fine_trainmode']) < 2e-4
    netMR.query(seeded_points(14, 600), calib_tensor())
    assert np.abs(netMR.get_preds()[0, 0].cpu().numpy() - gold['fine']).max() < 1e-4


def test_gen_mesh_accepts_rgbd_input(tmp_path):
    """The whole `gen_mesh` on a 6-channel subject (what bench.py runs by default): the mesh comes out and the strip saved
    beside it shows the colour half of the input next to the two predicted normal maps."""
    from pifuhd_amd import config, nets, reconstruction, synth
    oc = config.coarse_opt(in_image_channels=6)
    of = config.fine_opt(oc)
    of.loadSizeBig = 128
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=1, mlp_std=0.16, perturb=True)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = synth.synthetic_inputs(128, 6, seed=2)
    data['img_512'] = synth.synthetic_image(64, 6, seed=2)
    netMR.filter_global(data['img_512']); netMR.filter_local(data['img'][:, None])
    synth.calibrate_occupancy_bias(netMR, data['calib'], sd, 0.3, resolution=32)
    verts, faces, color = reconstruction.gen_mesh(64, netMR, 'cuda', data, str(tmp_path / 'rgbd.obj'))
    assert len(verts) > 100 and len(faces) > 100 and color.shape == verts.shape
    from PIL import Image
    assert Image.open(tmp_path / 'rgbd.png').size == (3 * 64, 64)


def test_reloading_weights_into_a_live_net_refreshes_every_derived_buffer():
    """State dict A, filter, then state dict B into the SAME nets: the outputs must equal those of fresh nets loaded
    with B (folded eval-mode BatchNorm scale / shift, packed conv weights and the repacked 7x7 output conv are all
    derived caches inside the context)."""
    from pifuhd_amd import nets, synth
    sd_a, oc, of = golden_nets(seed=1)
    sd_b, _, _ = golden_nets(seed=2)
    of.loadSizeBig = 128
    img64, img128 = synth.synthetic_image(64, 3, seed=2), synth.synthetic_image(128, 3, seed=3)

    def run(netG, netMR):
        netMR.filter_global(img64)
        netMR.filter_local(img128[:, None])
        return [t.clone() for t in (netG.nmlF, netG.nmlB, netG.im_feat_list[0], netMR.im_feat_list[0])]
    netG, netMR = nets.build_nets(oc, of, sd_a, 'cuda')
    first = run(netG, netMR)
    netMR.load_state_dict(sd_b)
    again = run(netG, netMR)
    fresh = run(*nets.build_nets(oc, of, sd_b, 'cuda'))
    assert all(torch.equal(a, b) for a, b in zip(again, fresh))
    assert not torch.equal(first[2], again[2])
    with pytest.raises(RuntimeError, match='Unexpected key'):
        netMR.load_state_dict({**sd_b, 'netG.mlp.filters.9.weight': torch.zeros(1)})
    with pytest.raises(RuntimeError, match='Missing key'):
        netMR.load_state_dict({k: v for k, v in sd_b.items() if k != 'mlp.filters.0.bias'})


def test_full_size_encoders_to_occupancy_end_to_end(golden_dir):
    """The path END TO END at full size against the reference's own run of it (`tests/golden/make_golden_r3.py e2e_full`):
    `filter_global` on the 512^2 image, `filter_local` on the 1024^2 image, then `query` on 4 096 seeded points - the
    occupancy the real encoders' feature maps lead to, not features compared on their own - within the north star's 1e-4;
    `calc_normal` on the first 1 024 points.  (`PIFuMRNet.py:71-186`, `:188-243`.)"""
    from helpers import seeded_points
    from pifuhd_amd import config, nets, synth
    gold = np.load(os.path.join(golden_dir, 'e2e_full.npz'))
    seed, n, lim = int(gold['seed_points_lim'][0]), int(gold['seed_points_lim'][1]), float(gold['seed_points_lim'][2])
    oc = config.coarse_opt(); of = config.fine_opt(oc)
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
    last = sorted(k for k in sd if k.startswith('mlp.filters.') and k.endswith('.bias'))[-1]
    sd[last] = sd[last] + float(gold['fine_last_bias_key_shift'][0])
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = synth.synthetic_inputs(1024, 3, seed=0)
    netMR.filter_global(data['img_512'])
    netMR.filter_local(data['img'][:, None])
    pts = seeded_points(seed, n, lim=lim)
    calib = data['calib']
    netMR.query(pts[:, None], calib_local=calib[:, None], calib_global=calib)
    preds = netMR.get_preds().cpu().numpy().reshape(-1)
    want = gold['preds']
    assert 0.2 < (want > 0.5).mean() < 0.8                               # the sample straddles the surface
    err = np.abs(preds - want).max()
    print('end-to-end occupancy: max abs diff %.2e (coarse %.2e)' %
          (err, np.abs(netG.get_preds().cpu().numpy().reshape(-1) - gold['preds_low']).max()))
    assert err < 1e-4
    assert np.abs(netG.get_preds().cpu().numpy().reshape(-1) - gold['preds_low']).max() < 1e-4
    assert _rel(netG.phi[0, ::8, ::16].cpu().numpy(), gold['phi_sub']) < 5e-4
    m = int(gold['nmls'].shape[1])
    netMR.calc_normal(pts[:, None, :, :m], calib[:, None], calib)
    got_n, want_n = netMR.nmls.cpu().numpy()[0], gold['nmls']
    # unit normals from finite differences of occupancies that agree to ~1e-6: compare directions where the gradient
    # is not vanishing (the reference divides by |g| + 1e-8)
    cos = (got_n * want_n).sum(0)
    assert np.median(cos) > 0.999 and (cos > 0.9).mean() > 0.97
    # the same two calls through the pixel tables (forced on at this size; gen_mesh's calc_normal batches take them by
    # default): same goldens of the reference's run, on the feature map the real encoder produced
    from pifuhd_amd import _lib
    _lib.call('pifu_set_pixel_tables', netMR._ctx.h, _lib.PIX_ALWAYS)
    try:
        netMR.query(pts[:, None], calib_local=calib[:, None], calib_global=calib)
        preds_px = netMR.get_preds().cpu().numpy().reshape(-1)
        low_px = netG.get_preds().cpu().numpy().reshape(-1)
        netMR.calc_normal(pts[:, None, :, :m], calib[:, None], calib)
        cos_px = (netMR.nmls.cpu().numpy()[0] * want_n).sum(0)
    finally:
        _lib.call('pifu_set_pixel_tables', netMR._ctx.h, _lib.PIX_AUTO)
    print('through the pixel tables: max abs diff %.2e (coarse %.2e), %.2e from the per-point path' %
          (np.abs(preds_px - want).max(), np.abs(low_px - gold['preds_low']).max(), np.abs(preds_px - preds).max()))
    assert not np.array_equal(preds_px, preds)
    assert np.abs(preds_px - want).max() < 1e-4 and np.abs(low_px - gold['preds_low']).max() < 1e-4
    assert np.median(cos_px) > 0.999 and (cos_px > 0.9).mean() > 0.97


def test_full_size_pifuhd_checkpoint_configuration_end_to_end(golden_dir, observed):
    """The configuration of the published PIFuHD checkpoint (`MLP.py:37-41`: `mlp_norm` none builds no norm
    layers; `Filter.py:37-46`: GroupNorm(32, C) in both HGFilters), END TO END at full size against the reference's own run
    (`tests/golden/make_golden_r5.py`): `filter_global(512^2)` + `filter_local(1024^2)` + `query` on 4 096 points +
    `calc_normal`.  The un-normalised MLP amplifies fp32 summation-order noise, so besides the north star's 1e-4
    the occupancy is held to a multiple of the CPU oracle's own distance from the same golden - and both are recorded."""
    from helpers import seeded_points
    from oracle import pifu_oracle as O
    from pifuhd_amd import config, nets, synth
    gold = np.load(os.path.join(golden_dir, 'e2e_full_pifuhd.npz'))
    seed, n, lim = int(gold['seed_points_lim'][0]), int(gold['seed_points_lim'][1]), float(gold['seed_points_lim'][2])
    oc = config.coarse_opt(mlp_norm='no', norm='group'); of = config.fine_opt(oc)
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=float(gold['mlp_std'][0]), perturb=False)
    assert not any('.norms.' in k for k in sd if k.startswith(('mlp.', 'netG.mlp.')))
    last = sorted(k for k in sd if k.startswith('mlp.filters.') and k.endswith('.bias'))[-1]
    sd[last] = sd[last] + float(gold['fine_last_bias_key_shift'][0])
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = synth.synthetic_inputs(1024, 3, seed=0)
    netMR.filter_global(data['img_512'])
    netMR.filter_local(data['img'][:, None])
    fc, ff = netG.im_feat_list[-1], netMR.im_feat_list[-1]
    e_fc = _rel(fc[0, ::16, ::8, ::8].cpu().numpy(), gold['feat_coarse_sub'])
    e_ff = _rel(ff[0, ::4, ::32, ::32].cpu().numpy(), gold['feat_fine_sub'])
    assert e_fc < 5e-4 and e_ff < 5e-4, (e_fc, e_ff)
    pts = seeded_points(seed, n, lim=lim)
    calib = data['calib']
    # the oracle on the GPU's feature maps: what fp32 CPU arithmetic leaves on this case
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    ref.feat_coarse, ref.feat_fine = fc.cpu(), ff.cpu()
    with torch.no_grad():
        o_fine = ref.query(pts, calib)[0, 0].numpy()
        o_low = ref.preds_coarse[0, 0].numpy()
    want, want_low = gold['preds'], gold['preds_low']
    assert 0.2 < (want > 0.5).mean() < 0.8 and want.max() - want[want > 0].min() > 0.5     # a field with contrast
    eo, eo_low = float(np.abs(o_fine - want).max()), float(np.abs(o_low - want_low).max())
    netMR.query(pts[:, None], calib_local=calib[:, None], calib_global=calib)
    preds, low = netMR.get_preds().cpu().numpy().reshape(-1), netG.get_preds().cpu().numpy().reshape(-1)
    eg, eg_low = float(np.abs(preds - want).max()), float(np.abs(low - want_low).max())
    from pifuhd_amd import _lib
    _lib.call('pifu_set_pixel_tables', netMR._ctx.h, _lib.PIX_ALWAYS)
    try:
        netMR.query(pts[:, None], calib_local=calib[:, None], calib_global=calib)
        eg_px = float(np.abs(netMR.get_preds().cpu().numpy().reshape(-1) - want).max())
    finally:
        _lib.call('pifu_set_pixel_tables', netMR._ctx.h, _lib.PIX_AUTO)
    observed('e2e_full_pifuhd', gpu_fine=eg, gpu_coarse=eg_low, gpu_fine_pixel_tables=eg_px, oracle_fine=eo, oracle_coarse=eo_low,
             feat_coarse_rel=e_fc, feat_fine_rel=e_ff)
    assert np.array_equal(preds == 0, want == 0)
    # north star 1e-4; and never more than a few times what the fp32 CPU oracle leaves from the same feature maps
    for e, o in ((eg, eo), (eg_low, eo_low), (eg_px, eo)):
        assert e < 1e-4 and e <= max(2e-5, 4 * o), (eg, eg_low, eg_px, eo, eo_low)
    m = int(gold['nmls'].shape[1])
    netMR.calc_normal(pts[:, None, :, :m], calib[:, None], calib)
    cos = (netMR.nmls.cpu().numpy()[0] * gold['nmls']).sum(0)
    assert np.median(cos) > 0.999 and (cos > 0.9).mean() > 0.95


def test_winograd_trunk_against_the_direct_convolutions(observed):
    """The ResnetBlock trunks of the normal nets (`networks.py:162-207`: 18 convolutions 1024 -> 1024, 3 x 3, reflection pad,
    on the 32 x 32 map of a 512 x 512 image) run as Winograd F(2x2, 3x3) - 16 batched GEMMs + one channel-separable bridge
    kernel between two of them.  Against the same nets with `PIFU_ENC_NO_WINOGRAD` (direct convolutions, the round-4 path): the
    normal maps agree to fp32 rounding, are not bit-identical (both paths ran), and the full-size goldens of the reference hold
    for both (`test_full_size_*`)."""
    from pifuhd_amd import config, nets, synth
    oc = config.coarse_opt(); of = config.fine_opt(oc)
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    img = synth.synthetic_inputs(1024, 3, seed=0)['img_512']
    netG.filter(img)
    wino = [t.clone() for t in (netG.nmlF, netG.nmlB, netG.im_feat_list[0])]
    netG._direct_trunk = True
    netG.filter(img)
    direct = [t.clone() for t in (netG.nmlF, netG.nmlB, netG.im_feat_list[0])]
    netG._direct_trunk = False
    netG.filter(img)
    again = [netG.nmlF, netG.nmlB, netG.im_feat_list[0]]
    errs = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(wino, direct)]
    observed('winograd_vs_direct', nmlF_rel=errs[0], nmlB_rel=errs[1], feat_coarse_rel=errs[2])
    assert all(not torch.equal(a, b) for a, b in zip(wino[:2], direct[:2]))
    # observed: 1.9e-5 on the normal maps (18 trunk convolutions deep, behind a tanh), 1.5e-6 on the coarse features they feed
    assert max(errs[:2]) < 1e-4 and errs[2] < 2e-5, errs
    assert all(torch.equal(a, b) for a, b in zip(wino, again))          # deterministic, and the switch leaves nothing behind


def test_overlapped_encoder_schedule_is_bit_identical():
    """`reconstruction._Subject` issues the two filter calls so that independent work overlaps (netF beside netB inside the
    coarse call, the coarse hourglass - left running by PIFU_ENC_DEFER_JOIN - beside the fine level, the fine image's upload
    beside the normal nets).  Only the schedule changes: every output must be bit-identical to the plain
    `filter_global(); filter_local()` sequence, call after call, and a query issued right after must see both maps."""
    from pifuhd_amd import config, nets, reconstruction, synth
    from helpers import seeded_points
    oc = config.coarse_opt(); of = config.fine_opt(oc)
    of.loadSizeBig = 256
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    data = {'name': 't', 'img': synth.synthetic_image(256, 3, seed=3), 'img_512': synth.synthetic_image(128, 3, seed=2),
            'calib': synth.synthetic_inputs(512, 3)['calib'], 'b_min': [-1, -1, -1], 'b_max': [1, 1, 1]}
    pts = seeded_points(5, 3000)
    calib = data['calib']
    netMR.filter_global(data['img_512'])
    netMR.filter_local(data['img'][:, None])
    torch.cuda.synchronize()
    want = [t.clone() for t in (netG.nmlF, netG.nmlB, netG.im_feat_list[0], netMR.im_feat_list[0])]
    netMR.query(pts[:, None], calib_local=calib[:, None], calib_global=calib)
    want_pred = netMR.get_preds().clone()
    for _ in range(3):
        reconstruction._Subject(netMR, data)
        netMR.query(pts[:, None], calib_local=calib[:, None], calib_global=calib)      # joins the deferred hourglass itself
        assert torch.equal(netMR.get_preds(), want_pred)
        got = [netG.nmlF, netG.nmlB, netG.im_feat_list[0], netMR.im_feat_list[0]]
        assert all(torch.equal(a, b) for a, b in zip(got, want))
    # a deferred coarse call followed directly by a coarse-only query (no filter_local in between)
    netG.filter(data['img_512'], _defer=True)
    netG.query(pts, calib)
    assert torch.equal(netG.im_feat_list[0], want[2])


def test_deferred_hourglass_is_joined_by_every_stream():
    """`pifu_filter_coarse(PIFU_ENC_DEFER_JOIN)` leaves the hourglass running on a stream of the library.  Every stream that
    asks for its result while it is still running must wait for it - not only the first one (`pifu_enc_join`): a query
    issued on a second stream right after another stream made the first call, with no synchronisation in between, sees the
    finished map; a call that converts its input (float64 host image -> float32 device copy, a temporary of `filter()`)
    stays correct because the net holds the converted tensor until the next filter call."""
    from pifuhd_amd import config, nets, synth
    from helpers import seeded_points
    oc = config.coarse_opt(); of = config.fine_opt(oc)
    of.loadSizeBig = 256
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=0, mlp_std=0.16)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    img = synth.synthetic_image(128, 3, seed=2)
    pts = seeded_points(5, 3000)
    calib = synth.synthetic_inputs(512, 3)['calib']
    netG.filter(img)
    netG.query(pts, calib)
    torch.cuda.synchronize()
    want_feat, want_pred = netG.im_feat_list[0].clone(), netG.get_preds().clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    fine_map = torch.randn(1, 16, 64, 64)
    for _ in range(3):
        netG.filter(img.double(), _defer=True)              # the converted copy is a temporary of filter()
        held = netG._coarse_in
        assert held.dtype == torch.float32 and held.is_cuda
        junk = [torch.full_like(held, float('nan')) for _ in range(4)]      # would recycle a freed input block
        with torch.cuda.stream(s1):                          # a first caller on one stream (it does not touch the coarse map) ...
            netMR.set_features(fine_map)
        with torch.cuda.stream(s2):                          # ... and the query on ANOTHER stream, which nobody ordered after
            netG.query(pts, calib)                           # the hourglass: it has to wait for it as well
            pred = netG.get_preds()
        torch.cuda.synchronize()
        del junk
        assert torch.equal(netG.im_feat_list[0], want_feat)
        assert torch.equal(pred, want_pred)


@pytest.mark.parametrize('front,back', [(True, False), (False, True), (False, False)])
def test_normal_map_option_variants(front, back, observed):
    """`use_front_normal` / `use_back_normal` (`options.py:162-163`, read from the checkpoint's opt: `PIFuNetwNML.py:33-36,64-67`,
    `PIFuMRNet.py:31-34,92-95`) switch a normal net and three input channels of both filters off.  Tiny sizes (64 / 128 px),
    both levels against the CPU oracle."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import config, nets, synth
    oc = config.coarse_opt(use_front_normal=front, use_back_normal=back)
    of = config.fine_opt(oc, loadSizeBig=128)
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=1, mlp_std=0.16, perturb=True)
    assert ('netG.netF.model.1.weight' in sd) == front and ('netG.netB.model.1.weight' in sd) == back
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    img64, img128 = synth.synthetic_image(64, 3, seed=2), synth.synthetic_image(128, 3, seed=3)
    netMR.filter_global(img64)
    netMR.filter_local(img128[:, None])
    assert (netG.nmlF is not None) == front and (netG.nmlB is not None) == back
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    with torch.no_grad():
        fc = ref.filter_global(img64).numpy()
        ff = ref.filter_local(img128).numpy()
    ec, ef = _rel(netG.im_feat_list[0].cpu().numpy(), fc), _rel(netMR.im_feat_list[0].cpu().numpy(), ff)
    observed(f'normal_map_options[front={front},back={back}]', coarse_rel=ec, fine_rel=ef)
    assert ec < 2e-4 and ef < 2e-4
    # and the occupancy through both levels
    calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
    pts = (torch.rand(1, 3, 3000, generator=torch.Generator().manual_seed(5)) * 2 - 1)
    netMR.query(pts, calib)
    with torch.no_grad():
        want = ref.query(pts, calib)[0, 0].numpy()
    got = netMR.get_preds()[0, 0].cpu().numpy()
    assert float(np.abs(got - want).max()) < 1e-4


@pytest.mark.parametrize('norm', ['batch', 'group'])
def test_other_filter_shapes(norm, observed):
    """Filter shapes other than the published ones (`options.py:86-93` are options): two stacks of depth-3 hourglasses with 128
    channels for the coarse level, 32 channels for the fine one; MLP input widths follow (129 / 288).  Tiny images, encoders and
    occupancy against the CPU oracle."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import config, nets, synth
    oc = config.coarse_opt(norm=norm, num_stack=2, hg_depth=3, hg_dim=128, mlp_dim=[129, 1024, 512, 256, 128, 1])
    of = config.fine_opt(oc, hg_dim=32, mlp_dim=[288, 512, 256, 128, 1], loadSizeBig=128)
    sd = synth.name_seeded_state_dict(synth.state_dict_spec(oc, of), seed=4, mlp_std=0.16, perturb=True)
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    img64, img128 = synth.synthetic_image(64, 3, seed=2), synth.synthetic_image(128, 3, seed=3)
    netMR.filter_global(img64)
    netMR.filter_local(img128[:, None])
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    with torch.no_grad():
        fc = ref.filter_global(img64).numpy()
        ff = ref.filter_local(img128).numpy()
    assert netG.im_feat_list[0].shape == fc.shape == (1, 128, 16, 16) and netMR.im_feat_list[0].shape == ff.shape == (1, 32, 64, 64)
    ec, ef = _rel(netG.im_feat_list[0].cpu().numpy(), fc), _rel(netMR.im_feat_list[0].cpu().numpy(), ff)
    observed(f'other_filter_shapes[{norm}]', coarse_rel=ec, fine_rel=ef)
    assert ec < 2e-4 and ef < 2e-4
    calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
    pts = (torch.rand(1, 3, 3000, generator=torch.Generator().manual_seed(5)) * 2 - 1)
    netMR.query(pts, calib)
    with torch.no_grad():
        want = ref.query(pts, calib)[0, 0].numpy()
    got = netMR.get_preds()[0, 0].cpu().numpy()
    assert float(np.abs(got - want).max()) < 1e-4


@pytest.mark.parametrize('coarse_px,fine_px', [(96, 192), (192, 384)])
def test_other_image_sizes(coarse_px, fine_px, observed):
    """Image sizes between the tiny and the full-size goldens, not powers of two (feature maps of 24 / 96 and 48 / 192 pixels:
    partly filled tiles in every convolution, odd split-K shapes, the normal maps upsampled by 2 with align_corners): both
    levels against the CPU oracle."""
    from oracle import pifu_oracle as O
    from pifuhd_amd import nets, synth
    sd, oc, of = golden_nets()
    of.loadSizeBig = fine_px
    netG, netMR = nets.build_nets(oc, of, sd, 'cuda')
    img_c, img_f = synth.synthetic_image(coarse_px, 3, seed=2), synth.synthetic_image(fine_px, 3, seed=3)
    netMR.filter_global(img_c)
    netMR.filter_local(img_f[:, None])
    ref = O.OracleNets(sd, oc, of, fine_training=True)
    with torch.no_grad():
        fc = ref.filter_global(img_c).numpy()
        ff = ref.filter_local(img_f).numpy()
    assert _rel(netG.nmlF.cpu().numpy(), ref.nmlF.numpy()) < 2e-4 and _rel(netG.nmlB.cpu().numpy(), ref.nmlB.numpy()) < 2e-4
    ec, ef = _rel(netG.im_feat_list[0].cpu().numpy(), fc), _rel(netMR.im_feat_list[0].cpu().numpy(), ff)
    observed(f'other_image_sizes[{coarse_px},{fine_px}]', coarse_rel=ec, fine_rel=ef)
    assert fc.shape == (1, 256, coarse_px // 4, coarse_px // 4) and ff.shape == (1, 16, fine_px // 2, fine_px // 2)
    assert ec < 5e-4 and ef < 5e-4
