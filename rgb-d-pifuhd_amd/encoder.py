"""Image encoders of the hot path: `filter()` / `filter_global()` / `filter_local()`.

INTERIM (round 1): the stacked-hourglass HGFilter (`Filter.py:23-228`) and the two pix2pixHD normal
generators (`networks.py:131-207`) are evaluated here with PyTorch-ROCm library ops ON THE GPU (MIOpen
convolutions), driven functionally from the reference's state-dict keys.  They run once per image
(~2 TFLOP) while the occupancy queries cost 9-390 TFLOP per mesh, so the hand-written HIP kernels went
to the query / grid / marching-cubes path first; replacing this file by MFMA implicit-GEMM kernels behind
`pifu_filter_*` entry points is the next row of DESIGN.md.  This is not a CPU fallback: tensors must live
on the GPU, and the module raises otherwise.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('encoder: input must be a GPU tensor (no CPU fallback in the product path)')


def _norm2d(sd, p, x, norm, bn_training):
    if norm == 'batch':
        if bn_training:      # batch statistics (SURVEY C-2: recon() leaves the fine net in train mode)
            return F.batch_norm(x, None, None, sd[p + '.weight'], sd[p + '.bias'], True, 0.1, 1e-5)
        return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                            False, 0.1, 1e-5)
    return F.group_norm(x, 32, sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def _conv_block(sd, p, x, norm, bt):
    cin = x.shape[1]
    o1 = F.conv2d(F.relu(_norm2d(sd, p + '.bn1', x, norm, bt)), sd[p + '.conv1.weight'], padding=1)
    o2 = F.conv2d(F.relu(_norm2d(sd, p + '.bn2', o1, norm, bt)), sd[p + '.conv2.weight'], padding=1)
    o3 = F.conv2d(F.relu(_norm2d(sd, p + '.bn3', o2, norm, bt)), sd[p + '.conv3.weight'], padding=1)
    cat = torch.cat([o1, o2, o3], 1)
    if cat.shape[1] != cin:
        return cat + F.conv2d(F.relu(_norm2d(sd, p + '.bn4', x, norm, bt)), sd[p + '.downsample.2.weight'])
    return cat + x


def _hourglass(sd, p, level, x, norm, bt):
    up1 = _conv_block(sd, f'{p}.b1_{level}', x, norm, bt)
    low = _conv_block(sd, f'{p}.b2_{level}', F.avg_pool2d(x, 2, stride=2), norm, bt)
    if level > 1:
        low = _hourglass(sd, p, level - 1, low, norm, bt)
    else:
        low = _conv_block(sd, f'{p}.b2_plus_{level}', low, norm, bt)
    low = _conv_block(sd, f'{p}.b3_{level}', low, norm, bt)
    return up1 + F.interpolate(low, scale_factor=2, mode='bicubic', align_corners=True)


def hg_filter(sd, p, x, n_stack, depth, norm, down_type, bn_training):
    """`Filter.forward` (`Filter.py:186-228`)."""
    x = F.relu(_norm2d(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], sd[p + '.conv1.bias'], stride=2, padding=3),
                       norm, bn_training))
    x = _conv_block(sd, p + '.conv2', x, norm, bn_training)
    if down_type == 'ave_pool':
        x = F.avg_pool2d(x, 2, stride=2)
    elif down_type != 'no_down':
        raise NameError('unknown downsampling type')          # Filter.py:197
    normx = x
    x = _conv_block(sd, p + '.conv3', x, norm, bn_training)
    x = _conv_block(sd, p + '.conv4', x, norm, bn_training)
    prev = x
    outs = []
    for s in range(n_stack):
        ll = _hourglass(sd, f'{p}.m{s}', depth, prev, norm, bn_training)
        ll = _conv_block(sd, f'{p}.top_m_{s}', ll, norm, bn_training)
        ll = F.conv2d(ll, sd[f'{p}.conv_last{s}.weight'], sd[f'{p}.conv_last{s}.bias'])
        ll = F.relu(_norm2d(sd, f'{p}.bn_end{s}', ll, norm, bn_training))
        out = F.conv2d(ll, sd[f'{p}.l{s}.weight'], sd[f'{p}.l{s}.bias'])
        outs.append(out)
        if s < n_stack - 1:
            prev = prev + F.conv2d(ll, sd[f'{p}.bl{s}.weight'], sd[f'{p}.bl{s}.bias']) \
                + F.conv2d(out, sd[f'{p}.al{s}.weight'], sd[f'{p}.al{s}.bias'])
    return outs, normx


def global_generator(sd, p, x, n_down=4, n_blocks=9):
    """`define_G(3,3,64,'global',4,9,1,3,'instance')` forward (`networks.py:131-207`)."""
    def inorm(t):
        return F.instance_norm(t, eps=1e-5)
    x = F.relu(inorm(F.conv2d(F.pad(x, (3, 3, 3, 3), mode='reflect'), sd[f'{p}.model.1.weight'], sd[f'{p}.model.1.bias'])))
    idx = 4
    for _ in range(n_down):
        x = F.relu(inorm(F.conv2d(x, sd[f'{p}.model.{idx}.weight'], sd[f'{p}.model.{idx}.bias'], stride=2, padding=1)))
        idx += 3
    for _ in range(n_blocks):
        q = f'{p}.model.{idx}.conv_block'
        h = F.relu(inorm(F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), sd[q + '.1.weight'], sd[q + '.1.bias'])))
        h = inorm(F.conv2d(F.pad(h, (1, 1, 1, 1), mode='reflect'), sd[q + '.5.weight'], sd[q + '.5.bias']))
        x = x + h
        idx += 1
    for _ in range(n_down):
        x = F.relu(inorm(F.conv_transpose2d(x, sd[f'{p}.model.{idx}.weight'], sd[f'{p}.model.{idx}.bias'], stride=2,
                                            padding=1, output_padding=1)))
        idx += 3
    x = F.conv2d(F.pad(x, (3, 3, 3, 3), mode='reflect'), sd[f'{p}.model.{idx + 1}.weight'], sd[f'{p}.model.{idx + 1}.bias'])
    return torch.tanh(x)


@torch.no_grad()
def filter_coarse(sd, opt, images, bn_training=False):
    """`PIFuNetwNML.filter` (`PIFuNetwNML.py:73-97`): returns (feature list, normx, nmlF, nmlB)."""
    _need_gpu(images)
    nmlF = nmlB = None
    nmls = []
    if getattr(opt, 'use_front_normal', False):
        nmlF = global_generator(sd, 'netF', images)
        nmls.append(nmlF)
    if getattr(opt, 'use_back_normal', False):
        nmlB = global_generator(sd, 'netB', images)
        nmls.append(nmlB)
    x = images
    if nmls:
        nm = torch.cat(nmls, 1)
        if images.shape[2:] != nm.shape[2:]:
            nm = F.interpolate(nm, size=images.shape[2:], mode='bilinear', align_corners=True)
        x = torch.cat([images, nm], 1)
    outs, normx = hg_filter(sd, 'image_filter', x, opt.num_stack, opt.hg_depth, opt.norm, opt.hg_down, bn_training)
    return outs, normx, nmlF, nmlB


@torch.no_grad()
def filter_fine(sd, opt, opt_coarse, images, nmlF, nmlB, bn_training=True):
    """`PIFuMRNet.filter_local` (`PIFuMRNet.py:83-118`), rect=None: images [1,C,H,W]."""
    _need_gpu(images)
    nmls = []
    if getattr(opt_coarse, 'use_front_normal', False):
        nmls.append(nmlF)
    if getattr(opt_coarse, 'use_back_normal', False):
        nmls.append(nmlB)
    x = images
    if nmls:
        big = F.interpolate(torch.cat(nmls, 1), size=(opt.loadSizeBig, opt.loadSizeBig), mode='bilinear',
                            align_corners=True)
        x = torch.cat([images, big], 1)
    outs, normx = hg_filter(sd, 'image_filter', x, opt.num_stack, opt.hg_depth, opt.norm, 'no_down', bn_training)
    return outs, normx
