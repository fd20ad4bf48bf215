"""The bench's body-like subject (`synth.craft_body_mlps`): the rewritten SurfaceClassifiers, evaluated by the CPU oracle exactly
as the reference would evaluate them, predict the analytic body the construction promises - |z| < 0.3 * H(x, y) - in the
GroupNorm configuration and in the un-normalised one of the published pifuhd.pt."""
import pytest
import torch

from oracle import pifu_oracle as O


@pytest.mark.parametrize('mlp_norm,mlp_std', [('group', 0.16), ('no', 0.10)])
def test_crafted_mlps_predict_the_analytic_body(mlp_norm, mlp_std):
    from pifuhd_amd import config, synth
    oc = config.coarse_opt(mlp_norm=mlp_norm)
    of = config.fine_opt(oc)
    spec = {}
    synth.mlp_keys('netG.mlp', oc.mlp_dim, oc.mlp_res_layers, oc.mlp_norm, spec)
    synth.mlp_keys('mlp', of.mlp_dim, of.mlp_res_layers, of.mlp_norm, spec)
    sd = synth.name_seeded_state_dict(spec, seed=0, mlp_std=mlp_std)
    g = torch.Generator().manual_seed(3)
    # a coarse feature map that carries the silhouette the way an encoder's does: smooth functions of the image + noise
    dome = synth.body_dome(128)
    lin = torch.linspace(-1, 1, 128)
    yy, xx = torch.meshgrid(lin, lin, indexing='ij')
    comps = torch.stack([dome, dome ** 2, xx, yy, xx * yy, torch.ones_like(xx)], 0)
    fc = torch.tanh(torch.einsum('ck,khw->chw', torch.randn(256, 6, generator=g), comps))[None] + 0.05 * torch.randn(1, 256, 128, 128, generator=g)
    ff = torch.randn(1, 16, 256, 256, generator=g)
    crafted, fit = synth.craft_body_mlps(sd, oc, of, fc)
    assert fit['fit_rms'] < 0.1 * fit['target_rms'] and set(crafted) == set(sd)
    # every weight outside group 0 of the rewritten layers (and the last fine layer) is the name-seeded one
    w_old, w_new = sd['netG.mlp.filters.1.weight'], crafted['netG.mlp.filters.1.weight']
    first_free = 512 // 32 if mlp_norm == 'group' else 4
    assert torch.equal(w_old[first_free:], w_new[first_free:]) and not torch.equal(w_old[:4], w_new[:4])
    ref = O.OracleNets(crafted, oc, of, fine_training=True)
    ref.feat_coarse, ref.feat_fine = fc, ff
    calib = torch.eye(4)[None]
    calib[0, 1, 1] = -1.0
    pts = torch.rand(1, 3, 6000, generator=g) * 2 - 1
    with torch.no_grad():
        # three reference-sized calls: under GroupNorm the chunk statistics differ from call to call, the body does not
        p = torch.cat([ref.query(pts[:, :, s:s + 2000], calib)[0, 0] for s in range(0, 6000, 2000)])
    xyz = O.orthogonal(pts, calib)
    u = 1.25 * O.index(dome[None, None], xyz[:, :2])[0, 0] - 0.25 - xyz[0, 2].abs() / 0.3
    wrong = (p > 0.5) != (u > 0)
    assert int((wrong & (u.abs() > 0.05)).sum()) == 0 and float(wrong.float().mean()) < 0.005
    assert 0.03 < float((u > 0).float().mean()) < 0.10
    # away from a shell around the surface the occupancy is saturated, like a trained net's
    far = u.abs() > 0.35
    assert float(torch.minimum(p[far], 1 - p[far]).max()) < 1e-3
