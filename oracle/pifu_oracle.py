"""CPU oracle for the PIFuHD occupancy-query + meshing hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch-CPU / numpy, fp32 on the network side, float64 on
the grid side, exactly like the reference) of the algorithm in `/root/reference` for SURVEY.md
section 8(a) rows A1-A18.  It exists to check the HIP path; only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it, and the product package never does.

Parity pin: every function here is checked in this container against the imported reference
(`tests/golden/make_golden.py`, which imports `/root/reference` and writes the small fixtures under
`tests/golden/*.npz`) and, for marching cubes, against scikit-image 0.18.3 (`oracle/mc_lewiner.c`,
fixtures from `tests/golden/make_golden_mc.py` under /opt/conda python3.9).  `tests/test_oracle_*.py`
re-check the oracle against those committed fixtures on every run.

Weights are consumed as a flat dict keyed by the reference's state-dict names.
"""
from __future__ import annotations

import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))


# =============================================================================== A1-A3
def index(feat: torch.Tensor, uv: torch.Tensor) -> torch.Tensor:
    """Bilinear sample, align_corners=True, zero padding (`BasePIFuNet.py:11-23`).

    feat [B,C,H,W], uv [B,2,N] -> [B,C,N].  Restated tap by tap instead of calling grid_sample.
    """
    B, C, H, W = feat.shape
    x = (uv[:, 0] + 1.0) * 0.5 * (W - 1)
    y = (uv[:, 1] + 1.0) * 0.5 * (H - 1)
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    wx1 = x - x0
    wy1 = y - y0
    wx0 = 1.0 - wx1
    wy0 = 1.0 - wy1
    out = torch.zeros(B, C, uv.shape[2], dtype=feat.dtype)
    flat = feat.reshape(B, C, H * W)
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            xi = (x0 + dx).long()
            yi = (y0 + dy).long()
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            lin = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1))
            g = torch.gather(flat, 2, lin[:, None, :].expand(B, C, -1))
            out = out + g * (wx * wy * ok.to(feat.dtype))[:, None, :]
    return out


def orthogonal(points: torch.Tensor, calib: torch.Tensor) -> torch.Tensor:
    """xyz = R p + t (`BasePIFuNet.py:25-43`, transform=None).  points [B,3,N], calib [B,4,4]."""
    rot = calib[:, :3, :3]
    trans = calib[:, :3, 3:4]
    return trans + torch.bmm(rot, points)


def depth_feature(xyz: torch.Tensor, load_size: int, z_size: float) -> torch.Tensor:
    """`DepthNormalizer.py:17-24`: z * (loadSize // 2) / z_size."""
    return xyz[:, 2:3, :] * (load_size // 2) / z_size


# =============================================================================== A4
def group_norm_1d(y, gamma, beta, groups=32, eps=1e-5):
    """GroupNorm over (channels-in-group x N points) of a [B,C,N] tensor (`MLP.py:38-39`)."""
    B, C, N = y.shape
    g = y.reshape(B, groups, (C // groups) * N)
    mean = g.mean(dim=2, keepdim=True)
    var = g.var(dim=2, unbiased=False, keepdim=True)
    g = (g - mean) / torch.sqrt(var + eps)
    return g.reshape(B, C, N) * gamma[None, :, None] + beta[None, :, None]


def batch_norm_1d(y, sd, prefix, training, eps=1e-5):
    if training:
        mean = y.mean(dim=(0, 2), keepdim=True)
        var = y.var(dim=(0, 2), unbiased=False, keepdim=True)
    else:
        mean = sd[prefix + '.running_mean'][None, :, None]
        var = sd[prefix + '.running_var'][None, :, None]
    return (y - mean) / torch.sqrt(var + eps) * sd[prefix + '.weight'][None, :, None] \
        + sd[prefix + '.bias'][None, :, None]


def mlp_forward(sd, prefix, feature, n_layers, res_layers, merge_layer, norm='group',
                training=False):
    """`MLP.forward` (`MLP.py:43-75`): k=1 conv chain with input skip-concats, norm + LeakyReLU(0.01),
    sigmoid last; returns (pred, phi) with phi = activation after layer `merge_layer`."""
    y = feature
    phi = None
    for i in range(n_layers):
        w = sd[f'{prefix}.filters.{i}.weight'][:, :, 0]
        b = sd[f'{prefix}.filters.{i}.bias']
        x = y if i not in res_layers else torch.cat([y, feature], 1)
        y = torch.matmul(w, x) + b[None, :, None]
        if i != n_layers - 1:
            if norm == 'group':
                y = group_norm_1d(y, sd[f'{prefix}.norms.{i}.weight'], sd[f'{prefix}.norms.{i}.bias'])
            elif norm == 'batch':
                y = batch_norm_1d(y, sd, f'{prefix}.norms.{i}', training)
            y = torch.where(y > 0, y, y * 0.01)
        if i == merge_layer:
            phi = y.clone()
    return torch.sigmoid(y), phi


def effective_merge_layer(merge_layer, dims):
    return merge_layer if merge_layer > 0 else len(dims) // 2       # MLP.py:25


# =============================================================================== A5-A6, A17
class OracleNets:
    """Holds the state dict + feature maps; mirrors the call order filter -> query -> get_preds."""

    def __init__(self, sd, opt_coarse, opt_fine=None, fine_training=True, coarse_training=False):
        self.sd = sd
        self.oc = opt_coarse
        self.of = opt_fine
        self.fine_training = fine_training          # SURVEY C-2: recon() leaves netMR in train mode
        self.coarse_training = coarse_training      # train_full_pifu: netG in train mode keeps every stack (PIFuNetwNML.py:94-97)
        self.feat_coarse_stacks = None
        self.feat_coarse = None
        self.feat_fine = None
        self.nmlF = None
        self.nmlB = None
        self.phi = None
        self.preds_coarse = None
        self.preds = None

    # ---- A5 -------------------------------------------------------------------------------
    def query_coarse(self, points, calib, prefix='netG.'):
        oc = self.oc
        xyz = orthogonal(points, calib)
        xy = xyz[:, :2, :]
        inside = ((xyz >= -1) & (xyz <= 1)).all(dim=1, keepdim=True).float()     # PIFuNetwNML.py:115-117
        z_feat = depth_feature(xyz, oc.loadSize, oc.z_size)
        feat = torch.cat([index(self.feat_coarse, xy), z_feat], 1)               # :128-129
        dims = oc.mlp_dim
        pred, phi = mlp_forward(self.sd, prefix + 'mlp', feat, len(dims) - 1, oc.mlp_res_layers,
                                effective_merge_layer(oc.merge_layer, dims), oc.mlp_norm, False)
        self.phi = phi
        self.preds_coarse = inside * pred
        return self.preds_coarse, phi

    # ---- A6 -------------------------------------------------------------------------------
    def query(self, points, calib, mask_fine=True):
        of = self.of
        xyz = orthogonal(points, calib)
        xy = xyz[:, :2, :]
        inside = ((xy >= -1) & (xy <= 1)).all(dim=1, keepdim=True).float()       # PIFuMRNet.py:150-152
        _, phi = self.query_coarse(points, calib)
        feat = torch.cat([index(self.feat_fine, xy), phi], 1)                    # :170-171
        dims = of.mlp_dim
        pred, _ = mlp_forward(self.sd, 'mlp', feat, len(dims) - 1, of.mlp_res_layers,
                              effective_merge_layer(-1, dims), of.mlp_norm, self.fine_training)
        self.preds = inside * pred if mask_fine else pred
        return self.preds

    # ---- f4: supervised forward (PIFuMRNet.py:157-186,253-288; PIFuNetwNML.py:119-120) --------
    def query_supervised(self, points, calib, labels):
        """`query(points, calib, labels=labels)` + `get_error()` with CustomBCELoss: returns the attributes the
        reference leaves behind (labels, w, gamma, preds, preds_low) and Err(occ:fine)."""
        xyz = orthogonal(points, calib)
        xy = xyz[:, :2, :]
        inside = ((xy >= -1) & (xy <= 1)).all(dim=1, keepdim=True).float()
        preds = self.query(points, calib)
        newlabels = inside * labels                                                   # PIFuMRNet.py:158
        w = inside.size(2) / inside.view(inside.size(0), -1).sum(1)                   # :160
        gamma = 1 - newlabels.view(newlabels.size(0), -1).sum(1) / inside.view(inside.size(0), -1).sum(1)   # :161-162
        preds_low = self.preds_coarse[None]
        # get_error (PIFuMRNet.py:267-270) evaluates the loss on preds_low for every fine stack (sic)
        err = custom_bce_loss(preds_low[0], newlabels, gamma, w)
        xyz_in = ((xyz >= -1) & (xyz <= 1)).all(dim=1, keepdim=True).float()
        return {'preds': preds, 'preds_low': preds_low, 'labels': newlabels, 'w': w, 'gamma': gamma,
                'err_fine': err, 'coarse_labels': xyz_in * labels}

    def query_supervised_full_pifu(self, points, calib, labels):
        """`train_full_pifu`: `netG.query` on every kept stack (`PIFuNetwNML.py:124-134`), `preds_low` = their stack,
        `Err(occ)` = mean loss over the stacks, `Err(occ:fine)` over preds_interm's length reading preds_low
        (`PIFuMRNet.py:154-155,255-270`).  The fine level consumes the LAST stack's phi."""
        xyz = orthogonal(points, calib)
        inside = ((xyz[:, :2] >= -1) & (xyz[:, :2] <= 1)).all(dim=1, keepdim=True).float()
        lows = []
        last = self.feat_coarse
        for f in self.feat_coarse_stacks:
            self.feat_coarse = f
            lows.append(self.query_coarse(points, calib)[0])
        self.feat_coarse = last
        preds = self.query(points, calib)
        newlabels = inside * labels
        w = inside.size(2) / inside.view(inside.size(0), -1).sum(1)
        gamma = 1 - newlabels.view(newlabels.size(0), -1).sum(1) / inside.view(inside.size(0), -1).sum(1)
        preds_low = torch.stack(lows, 0)
        err_occ = sum(custom_bce_loss(p_, newlabels, gamma, w) for p_ in preds_low) / preds_low.size(0)
        err_fine = custom_bce_loss(preds_low[0], newlabels, gamma, w)
        return {'preds': preds, 'preds_low': preds_low, 'err_occ': err_occ, 'err_fine': err_fine}

    # ---- A17 ------------------------------------------------------------------------------
    def calc_normal(self, points, calib, delta=0.001):
        """Forward differences (`PIFuMRNet.py:188-243`); points [1,3,n] -> [1,3,n]."""
        n = points.shape[2]
        stack = [points.clone() for _ in range(4)]
        for a in range(3):
            stack[a + 1][:, a, :] += delta
        pts_all = torch.stack(stack, 3).reshape(points.shape[0], 3, 4 * n)       # interleaved per point
        pred = self.query(pts_all, calib, mask_fine=False).reshape(points.shape[0], 1, n, 4)
        d = torch.cat([pred[..., 1] - pred[..., 0], pred[..., 2] - pred[..., 0],
                       pred[..., 3] - pred[..., 0]], 1)
        return F.normalize(-d, dim=1, eps=1e-8)

    # ---- C-9: the coarse net's own calc_normal / get_error (PIFuNetwNML.py:181-243) ---------------
    def calc_normal_coarse(self, points, calib, delta=0.1, return_pred=False):
        """`PIFuNetwNML.calc_normal`: forward differences of the coarse MLP's UNMASKED output over the 4n interleaved
        points [p, p+dx, p+dy, p+dz] (one MLP call: one GroupNorm batch); points [1,3,n] -> `nml` [1,3,n]."""
        oc = self.oc
        n = points.shape[2]
        stack = [points.clone() for _ in range(4)]
        for a in range(3):
            stack[a + 1][:, a, :] += delta
        pts_all = torch.stack(stack, 3).reshape(points.shape[0], 3, 4 * n)
        xyz = orthogonal(pts_all, calib)
        feat = torch.cat([index(self.feat_coarse, xyz[:, :2, :]), depth_feature(xyz, oc.loadSize, oc.z_size)], 1)
        dims = oc.mlp_dim
        pred, _ = mlp_forward(self.sd, 'netG.mlp', feat, len(dims) - 1, oc.mlp_res_layers,
                              effective_merge_layer(oc.merge_layer, dims), oc.mlp_norm, False)
        p4 = pred.reshape(points.shape[0], 1, n, 4)
        d = torch.cat([p4[..., 1] - p4[..., 0], p4[..., 2] - p4[..., 0], p4[..., 3] - p4[..., 0]], 1)
        nml = F.normalize(-d, dim=1, eps=1e-8)
        return (nml, pred) if return_pred else nml

    def get_error_coarse(self, points, calib, labels, gamma):
        """`netG.query(points, calib, labels=labels)` + `netG.get_error(gamma)` with CustomBCELoss, eval mode (one stack):
        labels masked by the x-y-z box (`PIFuNetwNML.py:115-120`), the criterion with a scalar gamma (`:233-236`)."""
        xyz = orthogonal(points, calib)
        inside = ((xyz >= -1) & (xyz <= 1)).all(dim=1, keepdim=True).float()
        pred, _ = self.query_coarse(points, calib)
        masked = inside * labels
        return custom_bce_loss(pred, masked, torch.tensor([float(gamma)])), masked

    # ---- A7-A10 ---------------------------------------------------------------------------
    def filter_global(self, img_512):
        oc = self.oc
        imgs = [img_512]
        if oc.use_front_normal:
            self.nmlF = global_generator(self.sd, 'netG.netF', img_512)
            imgs.append(self.nmlF)
        if oc.use_back_normal:
            self.nmlB = global_generator(self.sd, 'netG.netB', img_512)
            imgs.append(self.nmlB)
        x = torch.cat(imgs, 1)
        outs, _ = hg_filter(self.sd, 'netG.image_filter', x, oc.num_stack, oc.hg_depth, oc.norm,
                            oc.hg_down, bn_training=self.coarse_training)
        self.feat_coarse = outs[-1]                      # eval mode keeps the last stack only
        self.feat_coarse_stacks = outs if self.coarse_training else None
        return self.feat_coarse

    def filter_local(self, img):
        of = self.of
        nm = [t for t in (self.nmlF, self.nmlB) if t is not None]
        x = img
        if nm:
            big = F.interpolate(torch.cat(nm, 1), size=(of.loadSizeBig, of.loadSizeBig),
                                mode='bilinear', align_corners=True)             # PIFuMRNet.py:99-100
            x = torch.cat([img, big], 1)
        outs, _ = hg_filter(self.sd, 'image_filter', x, of.num_stack, of.hg_depth, of.norm,
                            'no_down', bn_training=self.fine_training)
        self.feat_fine = outs[-1]
        return self.feat_fine


def custom_bce_loss(pred, gt, gamma, w=None, brock=False, fixed_gamma=None):
    """`CustomBCELoss.forward` (`net_util.py:45-66`)."""
    x_hat = torch.clamp(pred, 1e-5, 1.0 - 1e-5)
    g = gamma[:, None, None] if fixed_gamma is None else fixed_gamma
    if brock:
        x = 3.0 * gt - 1.0
        loss = -(g * x * torch.log(x_hat) + (1.0 - g) * (1.0 - x) * torch.log(1.0 - x_hat))
    else:
        loss = -(g * gt * torch.log(x_hat) + (1.0 - g) * (1.0 - gt) * torch.log(1.0 - x_hat))
    if w is not None:
        if w.dim() == 1:
            w = w[:, None, None]
        return (loss * w).mean()
    return loss.mean()


# =============================================================================== A7 (HGFilter)
def _norm2d(sd, prefix, x, norm, bn_training):
    if norm == 'batch':
        if bn_training:
            return F.batch_norm(x, None, None, sd[prefix + '.weight'], sd[prefix + '.bias'], True, 0.1, 1e-5)
        return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                            sd[prefix + '.weight'], sd[prefix + '.bias'], False, 0.1, 1e-5)
    return F.group_norm(x, 32, sd[prefix + '.weight'], sd[prefix + '.bias'], 1e-5)


def conv_block(sd, p, x, norm, bn_training):
    """Pre-activation residual block with 3 concatenated 3x3 convs (`Filter.py:23-69`)."""
    cin = x.shape[1]
    o1 = F.conv2d(F.relu(_norm2d(sd, p + '.bn1', x, norm, bn_training)), sd[p + '.conv1.weight'], padding=1)
    o2 = F.conv2d(F.relu(_norm2d(sd, p + '.bn2', o1, norm, bn_training)), sd[p + '.conv2.weight'], padding=1)
    o3 = F.conv2d(F.relu(_norm2d(sd, p + '.bn3', o2, norm, bn_training)), sd[p + '.conv3.weight'], padding=1)
    cat = torch.cat([o1, o2, o3], 1)
    if cat.shape[1] != cin:
        res = F.conv2d(F.relu(_norm2d(sd, p + '.bn4', x, norm, bn_training)), sd[p + '.downsample.2.weight'])
    else:
        res = x
    return cat + res


def hourglass(sd, p, level, x, norm, bn_training):
    up1 = conv_block(sd, f'{p}.b1_{level}', x, norm, bn_training)
    low = conv_block(sd, f'{p}.b2_{level}', F.avg_pool2d(x, 2, stride=2), norm, bn_training)
    if level > 1:
        low = hourglass(sd, p, level - 1, low, norm, bn_training)
    else:
        low = conv_block(sd, f'{p}.b2_plus_{level}', low, norm, bn_training)
    low = conv_block(sd, f'{p}.b3_{level}', low, norm, bn_training)
    up2 = F.interpolate(low, scale_factor=2, mode='bicubic', align_corners=True)     # Filter.py:125
    return up1 + up2


def hg_filter(sd, p, x, n_stack, depth, norm, down_type, bn_training):
    """`Filter.forward` (`Filter.py:186-228`), use_sigmoid=False."""
    x = F.relu(_norm2d(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], sd[p + '.conv1.bias'],
                                                 stride=2, padding=3), norm, bn_training))
    x = conv_block(sd, p + '.conv2', x, norm, bn_training)
    if down_type == 'ave_pool':
        x = F.avg_pool2d(x, 2, stride=2)
    elif down_type != 'no_down':
        raise NameError('unknown downsampling type')
    normx = x
    x = conv_block(sd, p + '.conv3', x, norm, bn_training)
    x = conv_block(sd, p + '.conv4', x, norm, bn_training)
    prev = x
    outs = []
    for s in range(n_stack):
        ll = hourglass(sd, f'{p}.m{s}', depth, prev, norm, bn_training)
        ll = conv_block(sd, f'{p}.top_m_{s}', ll, norm, bn_training)
        ll = F.conv2d(ll, sd[f'{p}.conv_last{s}.weight'], sd[f'{p}.conv_last{s}.bias'])
        ll = F.relu(_norm2d(sd, f'{p}.bn_end{s}', ll, norm, bn_training))
        out = F.conv2d(ll, sd[f'{p}.l{s}.weight'], sd[f'{p}.l{s}.bias'])
        outs.append(out)
        if s < n_stack - 1:
            prev = prev + F.conv2d(ll, sd[f'{p}.bl{s}.weight'], sd[f'{p}.bl{s}.bias']) \
                + F.conv2d(out, sd[f'{p}.al{s}.weight'], sd[f'{p}.al{s}.bias'])
    return outs, normx


# =============================================================================== A9 (normal nets)
def global_generator(sd, p, x, n_down=4, n_blocks=9):
    """pix2pixHD GlobalGenerator, InstanceNorm(affine=False) (`networks.py:131-207`)."""
    def inorm(t):
        return F.instance_norm(t, eps=1e-5)
    x = F.relu(inorm(F.conv2d(F.pad(x, (3, 3, 3, 3), mode='reflect'), sd[f'{p}.model.1.weight'],
                              sd[f'{p}.model.1.bias'])))
    idx = 4
    for _ in range(n_down):
        x = F.relu(inorm(F.conv2d(x, sd[f'{p}.model.{idx}.weight'], sd[f'{p}.model.{idx}.bias'],
                                  stride=2, padding=1)))
        idx += 3
    for _ in range(n_blocks):
        q = f'{p}.model.{idx}.conv_block'
        h = F.relu(inorm(F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), sd[q + '.1.weight'], sd[q + '.1.bias'])))
        h = inorm(F.conv2d(F.pad(h, (1, 1, 1, 1), mode='reflect'), sd[q + '.5.weight'], sd[q + '.5.bias']))
        x = x + h
        idx += 1
    for _ in range(n_down):
        x = F.relu(inorm(F.conv_transpose2d(x, sd[f'{p}.model.{idx}.weight'], sd[f'{p}.model.{idx}.bias'],
                                            stride=2, padding=1, output_padding=1)))
        idx += 3
    x = F.conv2d(F.pad(x, (3, 3, 3, 3), mode='reflect'), sd[f'{p}.model.{idx + 1}.weight'],
                 sd[f'{p}.model.{idx + 1}.bias'])
    return torch.tanh(x)


# =============================================================================== A11-A14
def grid_points(resolution: int, calib_inv: np.ndarray, flat_index: np.ndarray | None = None) -> np.ndarray:
    """World points of the C-ordered lattice: float64 `calib^-1 . (-1 + 2 i / R, ...)`
    (`mesh_util.py:12-38,59-65`).  Returns float64 [3, n]; the caller casts to float32 like
    `mesh_util.py:70`.  `flat_index` selects a subset (C-order flat indices)."""
    R = resolution
    if flat_index is None:
        flat_index = np.arange(R * R * R, dtype=np.int64)
    i = flat_index // (R * R)
    j = (flat_index // R) % R
    k = flat_index % R
    step = 2.0 / R
    g = np.stack([i * step + (-1.0), j * step + (-1.0), k * step + (-1.0), np.ones_like(i, dtype=np.float64)], 1)
    return np.matmul(g, calib_inv.T)[:, :3].T


def grid_points_rect(res, calib_inv: np.ndarray, flat_index: np.ndarray) -> np.ndarray:
    """Same as grid_points for a rectangular lattice res[0] x res[1] x res[2] (multi-GPU slabs)."""
    r0, r1, r2 = (int(r) for r in res)
    i = flat_index // (r1 * r2)
    j = (flat_index // r2) % r1
    k = flat_index % r2
    g = np.stack([i * (2.0 / r0) + (-1.0), j * (2.0 / r1) + (-1.0), k * (2.0 / r2) + (-1.0),
                  np.ones_like(i, dtype=np.float64)], 1)
    return np.matmul(g, calib_inv.T)[:, :3].T


def batch_eval(points: np.ndarray, eval_func, num_samples: int) -> np.ndarray:
    """Consecutive chunks of `num_samples` points, remainder last (`mesh_util.py:98-114`)."""
    n = points.shape[1]
    out = np.zeros(n)
    for s in range(0, n, num_samples):
        out[s:s + num_samples] = eval_func(points[:, s:s + num_samples])
    return out


def eval_grid(resolution, calib_inv, eval_func, num_samples):
    pts = grid_points(resolution, calib_inv)
    return batch_eval(pts, eval_func, num_samples).reshape(resolution, resolution, resolution)


def eval_grid_octree(resolution, calib_inv, eval_func, num_samples, init_resolution=64, threshold=0.05,
                     stats=None):
    """Coarse-to-fine evaluation with uniform-cell skipping (`mesh_util.py:124-187`), written in the
    order-free per-voxel form of SURVEY App. G (bit-exact with the reference's sequential fill)."""
    R = resolution
    sdf = np.zeros((R, R, R))
    notproc = np.zeros((R, R, R), dtype=bool)
    notproc[:-1, :-1, :-1] = True
    reso = R // init_resolution
    ar = np.arange(R)
    while reso > 0:
        on = (ar % reso) == 0
        lattice = on[:, None, None] & on[None, :, None] & on[None, None, :]
        test = lattice & notproc
        flat = np.flatnonzero(test.reshape(-1))
        vals = batch_eval(grid_points(R, calib_inv, flat), eval_func, num_samples)
        if stats is not None:
            stats.append((reso, int(flat.size)))
        sdf.reshape(-1)[flat] = vals
        notproc.reshape(-1)[flat] = False
        if reso <= 1:
            break
        g = np.arange(0, R, reso)
        n = g.size
        v = sdf[np.ix_(g, g, g)]
        corners = np.stack([v[a:n - 1 + a, b:n - 1 + b, c:n - 1 + c]
                            for a in (0, 1) for b in (0, 1) for c in (0, 1)], 0)
        vmin = corners.min(0)
        vmax = corners.max(0)
        mean = 0.5 * (vmin + vmax)
        centre = g[:-1] + reso // 2
        skip = ((vmax - vmin) < threshold) & notproc[np.ix_(centre, centre, centre)]
        # per-voxel gather: winner = lexicographically largest covering skip cell
        best = np.full((R, R, R), -1, dtype=np.int64)
        cell_id = np.arange((n - 1) ** 3, dtype=np.int64).reshape(n - 1, n - 1, n - 1)
        cand = np.where(skip, cell_id, -1)
        p = np.arange(R)
        c_hi = np.minimum(p // reso, n - 2)                   # cell floor(p/reso), clipped
        c_lo = p // reso - 1                                  # previous cell, valid if p % reso == 0
        lo_ok = ((p % reso) == 0) & (c_lo >= 0)
        hi_ok = (p // reso) <= (n - 2)
        # a voxel p is covered by cell c on one axis iff c*reso <= p <= c*reso + reso
        for ax_choice in range(8):
            sel = []
            ok = []
            for axis in range(3):
                if (ax_choice >> axis) & 1:
                    sel.append(np.clip(c_lo, 0, n - 2)); ok.append(lo_ok)
                else:
                    sel.append(c_hi); ok.append(hi_ok & (p <= c_hi * reso + reso))
            c = cand[np.ix_(sel[0], sel[1], sel[2])]
            m = ok[0][:, None, None] & ok[1][None, :, None] & ok[2][None, None, :]
            c = np.where(m, c, -1)
            best = np.maximum(best, c)
        hit = best >= 0
        sdf[hit] = mean.reshape(-1)[best[hit]]
        notproc[hit] = False
        reso //= 2
    return sdf


# =============================================================================== A15 (Lewiner MC)
_MC_LIB = None


def _mc_lib():
    global _MC_LIB
    if _MC_LIB is None:
        path = os.path.join(_HERE, 'libmc_oracle.so')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} missing: run `make -C oracle` (or __graft_entry__.build())')
        lib = ctypes.CDLL(path)
        lib.mc_lewiner_oracle.restype = ctypes.c_int
        lib.mc_lewiner_oracle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        _MC_LIB = lib
    return _MC_LIB


def marching_cubes_lewiner(volume: np.ndarray, level: float):
    """Serial Lewiner marching cubes with scikit-image 0.18.3 semantics (`mesh_util.py:84`; SURVEY
    App. B): returns (verts float32 [V,3] in array-axis order, faces int32 [F,3] 'descent' order)."""
    vol = np.ascontiguousarray(volume, dtype=np.float32)
    if vol.ndim != 3 or min(vol.shape) < 2:
        raise ValueError('Input array must be at least 2x2x2.')
    level = float(level)
    if level < vol.min() or level > vol.max():
        raise ValueError('Surface level must be within volume data range.')
    lib = _mc_lib()
    nv = ctypes.c_int64(0)
    nf = ctypes.c_int64(0)
    lib.mc_lewiner_oracle(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2], level,
                          None, 0, None, 0, ctypes.byref(nv), ctypes.byref(nf))
    if nv.value == 0:
        raise RuntimeError('No surface found at the given iso value.')
    verts = np.empty((nv.value, 3), dtype=np.float32)
    faces = np.empty((nf.value, 3), dtype=np.int32)
    rc = lib.mc_lewiner_oracle(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2], level,
                               verts.ctypes.data, nv.value, faces.ctypes.data, nf.value,
                               ctypes.byref(nv), ctypes.byref(nf))
    if rc != 0:
        raise RuntimeError(f'mc_lewiner_oracle failed: {rc}')
    return verts, faces


# =============================================================================== A12, A16, A18
def reconstruction(eval_func, calib: np.ndarray, resolution, thresh=0.5, use_octree=False,
                   num_samples=10000):
    """`mesh_util.reconstruction` (`mesh_util.py:40-96`): returns (verts f64, faces i32) or -1."""
    calib_inv = np.linalg.inv(calib)
    mat = np.eye(4)
    mat[0, 0] = mat[1, 1] = mat[2, 2] = 2.0 / resolution
    mat[:3, 3] = -1.0
    if use_octree:
        sdf = eval_grid_octree(resolution, calib_inv, eval_func, num_samples)
    else:
        sdf = eval_grid(resolution, calib_inv, eval_func, num_samples)
    try:
        verts, faces = marching_cubes_lewiner(sdf, thresh)
        trans = np.matmul(calib_inv, mat)
        verts = (np.matmul(trans[:3, :3], verts.T) + trans[:3, 3:4]).T
        if np.linalg.det(trans[:3, :3]) < 0.0:
            faces = faces[:, ::-1]
        return verts, faces, sdf
    except Exception:
        print('error cannot marching cubes')
        return -1


def obj_text(verts, faces, colors) -> str:
    """`save_obj_mesh_with_color` (`mesh_util.py:189-198`) as a string."""
    lines = ['v %.4f %.4f %.4f %.4f %.4f %.4f\n' % (v[0], v[1], v[2], c[0], c[1], c[2])
             for v, c in zip(verts, colors)]
    lines += ['f %d %d %d\n' % (f[0] + 1, f[2] + 1, f[1] + 1) for f in faces]
    return ''.join(lines)


def esti_color(color: np.ndarray, xyz: np.ndarray) -> np.ndarray:
    """`esti_color` (`reconstruction.py:180-229`): color uint16 [V,4], xyz float32 [V,3] (projected vertices).  Every vertex
    with z < 0 takes the mean colour of its (at most) ten nearest silhouette vertices (0 <= z < 0.001) on the left
    (x_s < x; |dy| ascending, then x descending) and ten on the right (x_s >= x; |dy| ascending, then x ascending); equal
    keys keep vertex order.  uint16 sums, float64 division, truncation on store."""
    xyz32 = np.asarray(xyz, dtype=np.float32)
    out = np.array(color, dtype=np.uint16, copy=True)
    z = xyz32[:, 2]
    sil = np.flatnonzero((z >= np.float32(0)) & (z < np.float32(0.001)))
    if len(sil) == 0:
        raise IndexError('no silhouette vertex')
    X = xyz32.astype(np.float64)
    for i in np.flatnonzero(z < np.float32(0)):
        dx = X[sil, 0] - X[i, 0]
        dy = np.abs(X[sil, 1] - X[i, 1])
        L, R = sil[dx < 0], sil[dx >= 0]
        dL, dR = dy[dx < 0], dy[dx >= 0]
        oL = np.lexsort((np.arange(len(L)), -X[L, 0], dL))[:10]            # last key is the primary one
        oR = np.lexsort((np.arange(len(R)), X[R, 0], dR))[:10]
        if len(oL) + len(oR) == 0:
            raise Exception('no colour source')
        total = np.zeros(4, dtype=np.uint16)
        for idx in list(L[oL]) + list(R[oR]):
            total = (total + out[idx]).astype(np.uint16)
        out[i] = (total / float(len(oL) + len(oR))).astype(np.uint16)
    return out


def face_components(faces: np.ndarray):
    """Edge-connected components of a triangle list (faces adjacent iff they share an edge used by exactly two faces):
    returns (label = smallest face index of the component, per face; open = the face has an edge not used by exactly
    two faces).  Plain union-find; the checker of `pifu_face_components`."""
    F = np.asarray(faces, dtype=np.int64)
    edges = {}
    for f, (a, b, c) in enumerate(F):
        for u, v in ((a, b), (b, c), (c, a)):
            edges.setdefault((min(u, v), max(u, v)), []).append(f)
    parent = list(range(len(F)))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    open_face = np.zeros(len(F), dtype=np.uint8)
    for fs in edges.values():
        if len(fs) == 2:
            a, b = find(fs[0]), find(fs[1])
            if a != b:
                parent[max(a, b)] = min(a, b)
        else:
            open_face[fs] = 1
    return np.array([find(f) for f in range(len(F))], dtype=np.int32), open_face


def merge_vertices(verts: np.ndarray, colours: np.ndarray, faces: np.ndarray):
    """What `trimesh.load(obj_path)` (default `process=True`, `reconstruction.py:332`) does to the vertex list before
    `split()` sees it - RESTATED from trimesh's published `merge_vertices` (trimesh is not installed here: parity unpinned):
    referenced vertices whose positions agree (rounded to 8 decimals; the OBJ holds 4) become one vertex, the first
    occurrence is kept with its colour, kept vertices stay in their original order, unreferenced vertices are dropped and
    the faces are re-indexed.  Faces are NOT filtered: triangles that become degenerate stay (`validate=False`).
    Returns (verts, colours, faces)."""
    F = np.asarray(faces, dtype=np.int64)
    key_of, order, remap = {}, [], np.full(len(verts), -1, dtype=np.int64)
    referenced = np.zeros(len(verts), dtype=bool)
    referenced[F.reshape(-1)] = True
    for i in np.flatnonzero(referenced):
        k = tuple(np.round(np.asarray(verts[i], dtype=np.float64) * 1e8).astype(np.int64))
        if k not in key_of:
            key_of[k] = len(order)
            order.append(i)
        remap[i] = key_of[k]
    order = np.asarray(order, dtype=np.int64)
    return verts[order], colours[order], remap[F].astype(np.int32)


def widest_watertight_component(verts: np.ndarray, faces: np.ndarray):
    """`mesh.split()` + the selection loop of `meshcleaning` (`reconstruction.py:333-342`), restated from trimesh's published
    behaviour: faces are adjacent iff they share an edge that exactly two faces of the mesh use; a component counts only
    if it is watertight ON ITS OWN (every edge of its faces is used exactly twice among ITS faces - an edge that three
    faces of the whole mesh share can still pair up inside one component); components are visited in the order of their
    smallest face index and the first one with the largest extent along axis 0 wins.  Returns the ascending face indices
    of that component; raises IndexError when no component qualifies (`cc[0]` on an empty list)."""
    F = np.asarray(faces, dtype=np.int64)
    label, _ = face_components(F)
    best, best_ext = None, None
    for r in np.unique(label):
        idx = np.flatnonzero(label == r)
        count = {}
        for f in idx:
            a, b, c = F[f]
            for u, v in ((a, b), (b, c), (c, a)):
                k = (min(u, v), max(u, v))
                count[k] = count.get(k, 0) + 1
        if any(n != 2 for n in count.values()):
            continue
        ext = np.ptp(verts[:, 0][F[idx]])
        if best is None or best_ext < ext:
            best, best_ext = idx, ext
    if best is None:
        raise IndexError('list index out of range')
    return best


def analytic_field(points: np.ndarray) -> np.ndarray:
    """Network-independent test field of SURVEY 8d: sigmoid((0.35 - |(0.8x,0.45y,1.6z)|) / 0.01)."""
    r = np.sqrt((0.8 * points[0]) ** 2 + (0.45 * points[1]) ** 2 + (1.6 * points[2]) ** 2)
    return (1.0 / (1.0 + np.exp(-(0.35 - r) / 0.01))).astype(np.float32).astype(np.float64)


# =============================================================================== f1: input pipeline (readData.py)
# PARITY UNPINNED for the cv2 part: OpenCV is a third-party dependency of the reference (unpinned version, not
# installed in this image, and its IPP-enabled wheels do not even agree with its own C++ path to the last bit).
# What is restated is the published C++ algorithm of cv::resize for CV_8UC3 / INTER_LINEAR (imgproc/resize.cpp:
# 11-bit fixed-point coefficients, HResizeLinear + VResizeLinear<uchar>, INTER_AREA fast path at exact 2x);
# `addrect`, the calibration matrices and ToTensor/Normalize follow readData.py line by line.
def addrect(img: np.ndarray, rect):
    """`addrect` (`readData.py:15-29`): crop rect = (x, y, w, h) with a zero border where it leaves the image."""
    x, y, w, h = [int(v) for v in rect]
    H, W = img.shape[:2]
    left = abs(x) if x < 0 else 0
    top = abs(y) if y < 0 else 0
    right = abs(W - (x + w)) if x + w >= W else 0
    bottom = abs(H - (y + h)) if y + h >= H else 0
    padded = np.zeros((H + top + bottom, W + left + right, img.shape[2]), img.dtype)
    padded[top:top + H, left:left + W] = img
    x, y = x + left, y + top
    return padded[y:y + h, x:x + w, :]


def _lin_coords(dsize, ssize, clamp_weights):
    scale = 1.0 / (float(dsize) / float(ssize))                  # cv::resize: scale = 1 / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_weights:                                            # x axis: fx = 0 at the borders (resize.cpp xofs/alpha loop)
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= ssize - 1
        f[hi] = 0; s[hi] = ssize - 1
        s0, s1 = s, np.minimum(s + 1, ssize - 1)
    else:                                                        # y axis: rows are clamped, weights kept
        s0, s1 = np.clip(s, 0, ssize - 1), np.clip(s + 1, 0, ssize - 1)
    w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)      # saturate_cast<short>: round half to even
    w1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s0, s1, w0, w1


def cv2_resize_linear_u8(img: np.ndarray, dsize: int) -> np.ndarray:
    """cv2.resize(img, (dsize, dsize)) for uint8 HxWxC, default INTER_LINEAR (C++ path, see the header above)."""
    H, W = img.shape[:2]
    src = img.astype(np.int64)
    if W == 2 * dsize and H == 2 * dsize:                        # INTER_LINEAR -> INTER_AREA fast path at exactly 2x
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _lin_coords(dsize, W, True)
    y0, y1, b0, b1 = _lin_coords(dsize, H, False)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # HResizeLinear (int)
    S0, S1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess_image(bgr: np.ndarray, rect, size: int) -> np.ndarray:
    """crop + resize + BGR->RGB + ToTensor + Normalize(0.5, 0.5) (`readData.py:68,83-84,96-103`) -> float32 [3,size,size]."""
    im = cv2_resize_linear_u8(addrect(bgr, rect), size)
    rgb = im[:, :, ::-1].astype(np.float32) / np.float32(255)
    return np.ascontiguousarray(((rgb - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1))


def calib_matrices(rect, w: int, h: int):
    """`calib` and `calib_world` of `readData.py:71-95` (float32 4x4 each)."""
    trans_mat = np.identity(4)
    scale_im2ndc = 1.0 / float(w // 2)
    scale = w / rect[2]
    trans_mat *= scale
    trans_mat[3, 3] = 1.0
    trans_mat[0, 3] = -scale * (rect[0] + rect[2] // 2 - w // 2) * scale_im2ndc
    trans_mat[1, 3] = scale * (rect[1] + rect[3] // 2 - h // 2) * scale_im2ndc
    intrinsic = np.matmul(trans_mat, np.identity(4))
    proj = np.identity(4)
    proj[1, 1] = -1
    return proj.astype(np.float32), intrinsic.astype(np.float32)
