"""Synthetic checkpoints and inputs for the hot path (no real checkpoint or image ships with the
reference: SURVEY.md section 8d).

* `state_dict_spec(opt_coarse, opt_fine)` enumerates the reference's state-dict keys and shapes for
  `PIFuMRNet(opt, netG=PIFuNetwNML(opt_netG))` (key scheme: `reconstruction.py:285-292`,
  `Filter.py:133-185`, `MLP.py:30-41`, `networks.py:131-207`).
* `name_seeded_state_dict(...)` fills every key from a generator seeded by FNV-1a64(key) ^ seed, so
  the reference (golden generator), the oracle and the HIP path all see identical tensors without a
  checkpoint file travelling anywhere.  Distribution mirrors `net_util.py:10-32` (conv ~ N(0, s),
  BatchNorm2d weight ~ N(1, 0.02)); `perturb=True` additionally randomises biases, affine norm
  parameters and BN running stats so that parity tests exercise those terms.
* `synthetic_inputs(...)` builds the `gen_mesh` data dict of `readData.py:106-116`.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

_FNV_OFFSET = 0xCBF29CE484222325
_FNV_PRIME = 0x100000001B3


def fnv1a64(text: str) -> int:
    h = _FNV_OFFSET
    for byte in text.encode('utf-8'):
        h ^= byte
        h = (h * _FNV_PRIME) & 0xFFFFFFFFFFFFFFFF
    return h


# ----------------------------------------------------------------------------- key/shape spec
def _norm2d_keys(prefix, ch, norm, out):
    out[prefix + '.weight'] = (ch,)
    out[prefix + '.bias'] = (ch,)
    if norm == 'batch':
        out[prefix + '.running_mean'] = (ch,)
        out[prefix + '.running_var'] = (ch,)
        out[prefix + '.num_batches_tracked'] = ()


def _convblock_keys(prefix, cin, cout, norm, out):
    out[prefix + '.conv1.weight'] = (cout // 2, cin, 3, 3)
    out[prefix + '.conv2.weight'] = (cout // 4, cout // 2, 3, 3)
    out[prefix + '.conv3.weight'] = (cout // 4, cout // 4, 3, 3)
    _norm2d_keys(prefix + '.bn1', cin, norm, out)
    _norm2d_keys(prefix + '.bn2', cout // 2, norm, out)
    _norm2d_keys(prefix + '.bn3', cout // 4, norm, out)
    _norm2d_keys(prefix + '.bn4', cin, norm, out)
    if cin != cout:
        _norm2d_keys(prefix + '.downsample.0', cin, norm, out)   # alias of bn4 (Filter.py:48-49)
        out[prefix + '.downsample.2.weight'] = (cout, cin, 1, 1)


def _hourglass_keys(prefix, level, feat, norm, out):
    _convblock_keys(f'{prefix}.b1_{level}', feat, feat, norm, out)
    _convblock_keys(f'{prefix}.b2_{level}', feat, feat, norm, out)
    if level > 1:
        _hourglass_keys(prefix, level - 1, feat, norm, out)
    else:
        _convblock_keys(f'{prefix}.b2_plus_{level}', feat, feat, norm, out)
    _convblock_keys(f'{prefix}.b3_{level}', feat, feat, norm, out)


def filter_keys(prefix, n_stack, depth, in_ch, last_ch, norm, down_type, out):
    """Keys of the reference `Filter` (HGFilter), `Filter.py:133-185`."""
    if down_type not in ('ave_pool', 'no_down'):
        raise NameError('unknown downsampling type')       # Filter.py:197
    out[prefix + '.conv1.weight'] = (64, in_ch, 7, 7)
    out[prefix + '.conv1.bias'] = (64,)
    _norm2d_keys(prefix + '.bn1', 64, norm, out)
    _convblock_keys(prefix + '.conv2', 64, 128, norm, out)
    _convblock_keys(prefix + '.conv3', 128, 128, norm, out)
    _convblock_keys(prefix + '.conv4', 128, 256, norm, out)
    for s in range(n_stack):
        _hourglass_keys(f'{prefix}.m{s}', depth, 256, norm, out)
        _convblock_keys(f'{prefix}.top_m_{s}', 256, 256, norm, out)
        out[f'{prefix}.conv_last{s}.weight'] = (256, 256, 1, 1)
        out[f'{prefix}.conv_last{s}.bias'] = (256,)
        _norm2d_keys(f'{prefix}.bn_end{s}', 256, norm, out)
        out[f'{prefix}.l{s}.weight'] = (last_ch, 256, 1, 1)
        out[f'{prefix}.l{s}.bias'] = (last_ch,)
        if s < n_stack - 1:
            out[f'{prefix}.bl{s}.weight'] = (256, 256, 1, 1)
            out[f'{prefix}.bl{s}.bias'] = (256,)
            out[f'{prefix}.al{s}.weight'] = (256, last_ch, 1, 1)
            out[f'{prefix}.al{s}.bias'] = (256,)


def mlp_keys(prefix, dims, res_layers, norm, out):
    """Keys of the reference `MLP` (SurfaceClassifier), `MLP.py:30-41`."""
    n = len(dims) - 1
    for i in range(n):
        cin = dims[i] + (dims[0] if i in res_layers else 0)
        out[f'{prefix}.filters.{i}.weight'] = (dims[i + 1], cin, 1)
        out[f'{prefix}.filters.{i}.bias'] = (dims[i + 1],)
    for i in range(n - 1):
        if norm == 'group':
            out[f'{prefix}.norms.{i}.weight'] = (dims[i + 1],)
            out[f'{prefix}.norms.{i}.bias'] = (dims[i + 1],)
        elif norm == 'batch':
            _norm2d_keys(f'{prefix}.norms.{i}', dims[i + 1], 'batch', out)


def global_generator_keys(prefix, in_nc, out_nc, ngf, n_down, n_blocks, out):
    """Keys of pix2pixHD `GlobalGenerator` with InstanceNorm(affine=False), `networks.py:131-165`."""
    out[f'{prefix}.model.1.weight'] = (ngf, in_nc, 7, 7)
    out[f'{prefix}.model.1.bias'] = (ngf,)
    idx = 4
    for i in range(n_down):
        c = ngf * 2 ** i
        out[f'{prefix}.model.{idx}.weight'] = (2 * c, c, 3, 3)
        out[f'{prefix}.model.{idx}.bias'] = (2 * c,)
        idx += 3
    c = ngf * 2 ** n_down
    for _ in range(n_blocks):
        for j in (1, 5):
            out[f'{prefix}.model.{idx}.conv_block.{j}.weight'] = (c, c, 3, 3)
            out[f'{prefix}.model.{idx}.conv_block.{j}.bias'] = (c,)
        idx += 1
    for i in range(n_down):
        c = ngf * 2 ** (n_down - i)
        out[f'{prefix}.model.{idx}.weight'] = (c, c // 2, 3, 3)      # ConvTranspose2d: (in, out, k, k)
        out[f'{prefix}.model.{idx}.bias'] = (c // 2,)
        idx += 3
    out[f'{prefix}.model.{idx + 1}.weight'] = (out_nc, ngf, 7, 7)
    out[f'{prefix}.model.{idx + 1}.bias'] = (out_nc,)


def image_filter_in_channels(opt_coarse) -> int:
    c = int(getattr(opt_coarse, 'in_image_channels', 3))
    if getattr(opt_coarse, 'use_front_normal', False):
        c += 3
    if getattr(opt_coarse, 'use_back_normal', False):
        c += 3
    return c


def coarse_spec(opt, prefix='') -> 'OrderedDict[str, tuple]':
    """State-dict spec of `PIFuNetwNML(opt)` (`PIFuNetwNML.py:31-67`)."""
    out = OrderedDict()
    p = prefix
    filter_keys(p + 'image_filter', opt.num_stack, opt.hg_depth, image_filter_in_channels(opt),
                opt.hg_dim, opt.norm, opt.hg_down, out)
    mlp_keys(p + 'mlp', opt.mlp_dim, opt.mlp_res_layers, opt.mlp_norm, out)
    cimg = int(getattr(opt, 'in_image_channels', 3))
    if getattr(opt, 'use_front_normal', False):
        global_generator_keys(p + 'netF', cimg, 3, 64, 4, 9, out)
    if getattr(opt, 'use_back_normal', False):
        global_generator_keys(p + 'netB', cimg, 3, 64, 4, 9, out)
    return out


def state_dict_spec(opt_coarse, opt_fine) -> 'OrderedDict[str, tuple]':
    """State-dict spec of `PIFuMRNet(opt_fine, netG=PIFuNetwNML(opt_coarse))` (`PIFuMRNet.py:29-55`)."""
    out = OrderedDict()
    filter_keys('image_filter', opt_fine.num_stack, opt_fine.hg_depth,
                image_filter_in_channels(opt_coarse), opt_fine.hg_dim, opt_fine.norm, 'no_down', out)
    mlp_keys('mlp', opt_fine.mlp_dim, opt_fine.mlp_res_layers, opt_fine.mlp_norm, out)
    out.update(coarse_spec(opt_coarse, 'netG.'))
    return out


# ----------------------------------------------------------------------------- values
def _is_norm_key(key: str) -> bool:
    parts = key.split('.')
    owner = parts[-2] if len(parts) >= 2 else ''
    return (owner.startswith('bn') or owner == '0' and 'downsample' in key or 'norms' in parts)


def name_seeded_tensor(key: str, shape, seed: int = 0, conv_std: float = 0.02,
                       mlp_std: float | None = None, perturb: bool = False) -> torch.Tensor:
    gen = torch.Generator().manual_seed((fnv1a64(key) ^ seed) & 0x7FFFFFFFFFFFFFFF)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros((), dtype=torch.int64)
    shape = tuple(shape)
    is_norm = _is_norm_key(key)
    is_mlp = '.filters.' in key or key.startswith('filters.')
    if leaf == 'weight' and not is_norm:
        std = mlp_std if (is_mlp and mlp_std is not None) else conv_std
        return torch.randn(shape, generator=gen) * std
    if leaf == 'weight' and is_norm:
        is_bn2d = 'norms' not in key.split('.')
        if is_bn2d or perturb:
            return 1.0 + 0.02 * torch.randn(shape, generator=gen) * (5.0 if perturb else 1.0)
        return torch.ones(shape)
    if leaf == 'bias':
        if perturb:
            return 0.05 * torch.randn(shape, generator=gen)
        return torch.zeros(shape)
    if leaf == 'running_mean':
        return 0.1 * torch.randn(shape, generator=gen) if perturb else torch.zeros(shape)
    if leaf == 'running_var':
        return 0.5 + torch.rand(shape, generator=gen) if perturb else torch.ones(shape)
    raise KeyError(f'unrecognised state-dict leaf in {key!r}')


def name_seeded_state_dict(spec, seed: int = 0, conv_std: float = 0.02, mlp_std: float | None = None,
                           perturb: bool = False) -> 'OrderedDict[str, torch.Tensor]':
    sd = OrderedDict()
    for key, shape in spec.items():
        sd[key] = name_seeded_tensor(key, shape, seed, conv_std, mlp_std, perturb)
    # the ConvBlock shortcut norm is the same module object as bn4 (Filter.py:41,48-49)
    for key in list(sd):
        if '.downsample.0.' in key:
            sd[key] = sd[key.replace('.downsample.0.', '.bn4.')]
    return sd


# ----------------------------------------------------------------------------- inputs
def synthetic_image(size: int, channels: int = 3, seed: int = 0) -> torch.Tensor:
    """[1, C, size, size] in [-1, 1]: low-frequency sinusoids + U(-0.1, 0.1) noise."""
    gen = torch.Generator().manual_seed(0x5EED0000 + seed)
    lin = torch.linspace(-1.0, 1.0, size)
    yy, xx = torch.meshgrid(lin, lin, indexing='ij')
    img = torch.empty(1, channels, size, size)
    for c in range(channels):
        fx, fy, ph = (1.5 + c + 0.37 * seed), (2.25 - 0.5 * c + 0.21 * seed), 0.7 * c + seed
        img[0, c] = 0.6 * torch.sin(math.pi * fx * xx + ph) * torch.cos(math.pi * fy * yy - 0.5 * ph)
        img[0, c] += 0.25 * torch.exp(-((xx - 0.1 * c) ** 2 + (yy + 0.05 * c) ** 2) * 4.0)
    img += (torch.rand(img.shape, generator=gen) - 0.5) * 0.2
    return img.clamp_(-1.0, 1.0)


_BODY_PARTS = [(0.0, 0.05, 0.36, 0.48), (0.0, -0.54, 0.17, 0.20), (-0.19, 0.64, 0.13, 0.30), (0.19, 0.64, 0.13, 0.30),
               (-0.42, -0.02, 0.12, 0.38), (0.42, -0.02, 0.12, 0.38)]        # (cx, cy, rx, ry): torso, head, legs, arms - overlapping


def body_dome(size: int, seed: int = 0) -> torch.Tensor:
    """[size, size] thickness profile of the synthetic body over [-1, 1]^2 (pixel centres at linspace(-1, 1, size), i.e. the
    align_corners sampling of `index()`): the union of six ellipsoids' half-heights, 1 on their axes, 0 outside the silhouette."""
    lin = torch.linspace(-1.0, 1.0, size)
    yy, xx = torch.meshgrid(lin, lin, indexing='ij')
    dome = torch.zeros(size, size)
    for k, (cx, cy, rx, ry) in enumerate(_BODY_PARTS):
        cx = cx + 0.02 * math.sin(1.7 * seed + k)
        r2 = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2
        dome = torch.maximum(dome, torch.clamp(1.0 - r2, min=0.0).sqrt())
    return dome


def synthetic_body_image(size: int, channels: int = 3, seed: int = 0) -> torch.Tensor:
    """[1, C, size, size] in [-1, 1]: a smooth upright silhouette (torso + head + limbs as soft-edged ellipses) on a black
    background, softly shaded, no pixel noise; channels 3.. of an RGB-D input carry a smooth dome-shaped depth inside the
    silhouette.  The input of the bench's body-like subject: what a photograph of a person in front of a masked background
    looks like to the encoders, as opposed to `synthetic_image`'s full-frame texture."""
    lin = torch.linspace(-1.0, 1.0, size)
    yy, xx = torch.meshgrid(lin, lin, indexing='ij')
    dome = body_dome(size, seed)
    mask = torch.clamp(dome * 6.0, max=1.0)                               # soft edge a few pixels wide
    img = torch.empty(1, channels, size, size)
    for c in range(channels):
        if c < 3:
            shade = 0.55 + 0.25 * dome + 0.08 * torch.sin(math.pi * (0.8 + 0.3 * c) * xx + 0.5 * c) * torch.cos(math.pi * 0.6 * yy)
        else:
            shade = 0.2 + 0.7 * dome                                      # depth-like channels
        img[0, c] = -1.0 + mask * (shade + 1.0)
    return img.clamp_(-1.0, 1.0)


def craft_body_mlps(sd, opt_coarse, opt_fine, feat_coarse, seed: int = 0, sharpness: float = 30.0, half_depth: float = 0.3,
                    ballast: float = 100.0, tail_std: float = 0.01):
    """Weights of the two SurfaceClassifiers that make the net predict ONE smooth closed body - what a trained PIFuHD does
    and name-seeded random weights do not (their 0.5 level set is a rough sheet of millions of vertices, or - at small weight
    scales - a field so flat that the octree never refines).  For synthetic checkpoints only; same architecture, same
    kernels, same FLOPs: a handful of "wire" channels carry the construction, every other weight stays as drawn.

    The body: |z| < half_depth * H(x, y), H = a linear read-out of the coarse feature map fitted (ridge regression, on the
    map the given encoders produce for this image) to `1.25 * body_dome - 0.25` - positive inside the silhouette, negative
    outside, ellipsoidal in profile.  The coarse MLP forms u = H - |z_n| / tau from z and the image features with two
    LeakyReLU layers (leaky(t) + leaky(-t) = 0.99 |t|), hands it to the fine MLP in phi, and the fine MLP carries it to the
    logit `sharpness * u` (u changes by 1 / (0.3 * 256) per voxel of a 512^3 grid along z: at the default 30 the occupancy
    goes from 0.05 to 0.95 over ~15 voxels and the octree of `reconstruction()` refines 2.3 % of the grid - the statistics
    SURVEY section 6 measured on a smooth body; at 100 it is a two-voxel step and 1.3 %).  Under GroupNorm (MLP.py:37-39)
    the wires sit in group 0 of every layer beside their own negatives (group mean exactly 0) and a +-`ballast` pair (the
    group's variance: the statistics of a chunk then scale the wires by 1 + O(1e-4), uniformly, which moves no zero of u).
    Returns (state dict, fit statistics).  `feat_coarse`: [1, C, H, W] as `netG.im_feat_list[-1]`."""
    oc, of = opt_coarse, opt_fine
    mode = oc.mlp_norm if oc.mlp_norm in ('group', 'batch') else 'no'
    if mode == 'batch' or list(oc.mlp_res_layers) != [2, 3, 4] or list(of.mlp_res_layers) != [1, 2] or int(oc.merge_layer) != 2 \
            or len(oc.mlp_dim) != 6 or len(of.mlp_dim) != 5:
        raise NotImplementedError('craft_body_mlps: the reference architecture with mlp_norm group / none')
    sd = OrderedDict((k, v.clone()) for k, v in sd.items())
    F = feat_coarse.detach().float().cpu()[0]                              # [C, h, w]
    C, h, w = F.shape
    if h != w or C + 1 != oc.mlp_dim[0]:
        raise ValueError('craft_body_mlps: square coarse map with mlp_dim[0] - 1 channels expected')
    target = (1.25 * body_dome(h, seed) - 0.25).reshape(-1).double()
    A = torch.cat([F.reshape(C, -1).T.double(), torch.ones(h * w, 1, dtype=torch.float64)], 1)
    lam = 1e-3 * float((A * A).mean())
    sol = torch.linalg.solve(A.T @ A + lam * torch.eye(C + 1, dtype=torch.float64), A.T @ target)
    fit = A @ sol
    v, c = sol[:C].float(), float(sol[C])
    stats = {'fit_rms': float((fit - target).pow(2).mean().sqrt()), 'target_rms': float(target.pow(2).mean().sqrt()),
             'inside_share_of_map': float((fit > 0).double().mean())}
    z_scale = float(oc.loadSize // 2) / float(oc.z_size)                   # DepthNormalizer: z_n = z * z_scale
    tau = half_depth * z_scale
    gn = mode == 'group'
    B = float(ballast)

    def layer(prefix, i, wires, n_out):
        """Rewrite group 0 of layer i: `wires` = [(row, {input column: weight}, bias)] - every wire followed by its negative in
        the next row -, then the ballast pair, the rest of the group silent; returns the norm's gain for the wires."""
        W, b = sd[f'{prefix}.filters.{i}.weight'], sd[f'{prefix}.filters.{i}.bias']
        group = n_out // 32 if gn else 2 * len(wires)
        W[:group] = 0.0
        b[:group] = 0.0
        r = 0
        for cols, bias in wires:
            for col, val in cols.items():
                W[r, col, 0] = val
                W[r + 1, col, 0] = -val
            b[r], b[r + 1] = bias, -bias
            r += 2
        if gn:
            if r + 2 > group:
                raise ValueError('craft_body_mlps: group 0 of %s layer %d is too small' % (prefix, i))
            b[r], b[r + 1] = B, -B
            g, be = sd[f'{prefix}.norms.{i}.weight'], sd[f'{prefix}.norms.{i}.bias']
            g[:group] = 1.0
            be[:group] = 0.0
            g[:r] = B * math.sqrt(2.0 / group)                            # = the group's sigma when the wires are small against B
    nc, nf = 'netG.mlp', 'mlp'
    d = oc.mlp_dim
    # coarse layer 0: +-z_n, +-H
    layer(nc, 0, [({C: 1.0}, 0.0), ({k: float(v[k]) for k in range(C)}, c)], d[1])
    # coarse layer 1: 0.99 |z_n| = leaky(z_n) + leaky(-z_n); 1.01 H = leaky(H) - leaky(-H)
    layer(nc, 1, [({0: 1.0, 1: 1.0}, 0.0), ({2: 1.0, 3: -1.0}, 0.0)], d[2])
    # coarse layer 2 (skip layer: inputs [previous | features]): u = H - |z_n| / tau  -> phi
    layer(nc, 2, [({2: 1.0 / 1.01 ** 2, 3: -1.0 / 1.01 ** 2, 0: -1.0 / (0.99 * tau)}, 0.0)], d[3])
    # fine MLP: inputs [fine features (16) | phi]; the wire pair leaky(u), leaky(-u) -> 1.01 u per layer
    cf = of.mlp_dim[0] - d[3]
    layer(nf, 0, [({cf + 0: 1.0, cf + 1: -1.0}, 0.0)], of.mlp_dim[1])
    layer(nf, 1, [({0: 1.0, 1: -1.0}, 0.0)], of.mlp_dim[2])
    layer(nf, 2, [({0: 1.0, 1: -1.0}, 0.0)], of.mlp_dim[3])
    last = len(of.mlp_dim) - 2
    gen = torch.Generator().manual_seed((fnv1a64('craft_body_tail') ^ seed) & 0x7FFFFFFFFFFFFFFF)
    Wl = sd[f'{nf}.filters.{last}.weight']
    # what the other channels add: a sub-voxel ripple (without norm layers their activations are several times larger than under
    # GroupNorm - the same weights would raise specks of a few faces beside the body - so the tail is scaled down there)
    Wl.copy_(torch.randn(Wl.shape, generator=gen) * (tail_std if gn else 0.3 * tail_std))
    if gn:
        Wl[0, :of.mlp_dim[3] // 32, 0] = 0.0                              # (the ballast pair of the last hidden layer)
    Wl[0, 0, 0], Wl[0, 1, 0] = sharpness / 1.01 ** 3, -sharpness / 1.01 ** 3
    sd[f'{nf}.filters.{last}.bias'].zero_()
    return sd, stats


def synthetic_inputs(load_size: int = 1024, channels: int = 3, seed: int = 0, kind: str = 'texture') -> dict:
    """Data dict with the keys `gen_mesh` reads (`readData.py:106-116`, `reconstruction.py:29-34`).  kind: 'texture' (the
    full-frame sinusoid + noise image) or 'body' (`synthetic_body_image`)."""
    img = synthetic_body_image(load_size, channels, seed) if kind == 'body' else synthetic_image(load_size, channels, seed)
    if load_size == 512:
        img_512 = img.clone()
    else:
        img_512 = torch.nn.functional.interpolate(img, size=(512, 512), mode='bilinear',
                                                  align_corners=False)
    calib = torch.eye(4)
    calib[1, 1] = -1.0                                   # readData.py:90-92
    return {
        'name': f'synthetic_{seed}',
        'img': img, 'img_512': img_512,
        'calib': calib.unsqueeze(0),
        'b_min': [-1, -1, -1], 'b_max': [1, 1, 1],
    }


def calibrate_occupancy_bias(net, calib_tensor, state_dict, frac_inside: float = 0.3, resolution: int = 64) -> float:
    """Random weights rarely put the 0.5 level set inside the volume.  For synthetic checkpoints only:
    shift the bias of the fine MLP's last layer so that `frac_inside` of a dense `resolution`^3 lattice is
    predicted inside (the shift does not touch any other activation: the layer is followed by the sigmoid
    only).  Updates `state_dict` in place, reloads the fine MLP and returns the shift."""
    from . import mesh_util
    sdf, _ = mesh_util.eval_grid_device(net, calib_tensor, resolution, 5000, use_octree=False)
    p = sdf.reshape(-1).clamp(1e-7, 1 - 1e-7)
    logit = torch.log(p / (1 - p))
    shift = -float(torch.quantile(logit.float()[::7], 1.0 - frac_inside))
    last = len(net.opt.mlp_dim) - 2
    key = f'mlp.filters.{last}.bias'
    state_dict[key] = state_dict[key] + shift
    net._sd[key] = net._sd[key] + shift
    net._push_weights()
    return shift
