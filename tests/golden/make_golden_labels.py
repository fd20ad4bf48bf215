#!/usr/bin/env python3
"""Golden fixture for the supervised forward: reference `PIFuMRNet.query(..., labels=...)`,
`get_error()` with `CustomBCELoss`, and `PIFuNetwNML.query(labels=...)`.  Imports /root/reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_labels.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG            # noqa: E402  (sets up the reference import stubs)
import numpy as np                  # noqa: E402
import torch                        # noqa: E402
from net_util import CustomBCELoss  # noqa: E402  (reference)


def main():
    out = {}
    netG, netMR, sd, oc, of = MG.build_reference('group', seed=1)
    netMR.criteria = {'occ': CustomBCELoss()}
    fc, ff = MG.seeded_features(3)
    netG.im_feat_list = [fc]
    netMR.im_feat_list = [ff]
    g = torch.Generator().manual_seed(77)
    n = 700
    pts = torch.rand(1, 3, n, generator=g) * 2.3 - 1.15            # some points outside the box
    labels = (torch.rand(1, 1, n, generator=g) > 0.5).float()
    calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
    with torch.no_grad():
        netMR.query(pts, calib, labels=labels)
        err = netMR.get_error()
    out['seed'] = np.array([77, n])
    out['preds'] = netMR.preds.numpy()
    out['preds_low'] = netMR.preds_low.numpy()
    out['labels'] = netMR.labels.numpy()
    out['w'] = netMR.w.numpy()
    out['gamma'] = netMR.gamma.numpy()
    out['err_fine'] = np.array([float(err['Err(occ:fine)'])])
    # coarse net alone
    with torch.no_grad():
        netG.query(pts, calib, labels=labels)
    out['coarse_labels'] = netG.labels.numpy()
    # the loss on its own (brock and plain), random operands
    pred = torch.rand(2, 1, 50, generator=g); gt = (torch.rand(2, 1, 50, generator=g) > 0.4).float()
    gam = torch.tensor([0.3, 0.6]); w = torch.tensor([1.5, 0.7])
    out['loss_in'] = np.stack([pred.numpy(), gt.numpy()])
    out['loss_plain'] = np.array([float(CustomBCELoss()(pred, gt, gam, w)), float(CustomBCELoss()(pred, gt, gam))])
    out['loss_brock'] = np.array([float(CustomBCELoss(brock=True)(pred, gt, gam, w)), float(CustomBCELoss(gamma=0.25)(pred, gt, gam, w))])
    np.savez_compressed(os.path.join(HERE, 'labels_small.npz'), **out)
    print({k: (v.shape, float(np.abs(v).mean())) for k, v in out.items()})


if __name__ == '__main__' and 'full' not in sys.argv[1:]:
    main()


def full_pifu():
    """`opt.train_full_pifu = True`: netG stays in TRAIN mode (`PIFuMRNet.py:59-69`), so its HGFilter
    normalises with batch statistics and keeps the outputs of ALL stacks (`PIFuNetwNML.py:94-97`); `netG.query` runs the
    MLP on every stack, `preds_low` stacks them and `get_error` adds `Err(occ)` over the stacks (`PIFuMRNet.py:255-262`).
    Tiny images (64 px coarse, 128 px fine)."""
    from pifuhd_amd import synth
    out = {}
    netG, netMR, sd, oc, of = MG.build_reference('group', seed=1)
    netG.opt.train_full_pifu = True
    netMR.opt.train_full_pifu = True
    netMR.opt.no_intermediate_loss = False
    netMR.opt.loadSizeBig = 128
    netMR.criteria = {'occ': CustomBCELoss()}
    netMR.train()
    netG.train()
    assert netG.training and netMR.training
    g = torch.Generator().manual_seed(78)
    n = 600
    pts = torch.rand(1, 3, n, generator=g) * 2.2 - 1.1
    labels = (torch.rand(1, 1, n, generator=g) > 0.5).float()
    calib = torch.eye(4)[None]; calib[0, 1, 1] = -1
    with torch.no_grad():
        netMR.filter_global(synth.synthetic_image(64, 3, seed=2))
        netMR.filter_local(synth.synthetic_image(128, 3, seed=3)[:, None])
        assert len(netG.im_feat_list) == 4
        netMR.query(pts, calib, labels=labels)
        err = netMR.get_error()
    out['seed'] = np.array([78, n])
    out['feat_stacks'] = torch.cat(netG.im_feat_list, 0).numpy()          # [4, 256, 16, 16]
    out['preds'] = netMR.preds.numpy()
    out['preds_low'] = netMR.preds_low.numpy()                              # [4, 1, 1, n]
    out['err_occ'] = np.array([float(err['Err(occ)'])])
    out['err_fine'] = np.array([float(err['Err(occ:fine)'])])
    np.savez_compressed(os.path.join(HERE, 'labels_full_pifu.npz'), **out)
    print({k: (v.shape, float(np.abs(v).mean())) for k, v in out.items()})


if __name__ == '__main__' and 'full' in sys.argv[1:]:
    full_pifu()
