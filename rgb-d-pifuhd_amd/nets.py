"""Host-side mirrors of the reference's network classes for the hot path.

`PIFuNetwNML` (coarse level + normal nets, reference `PIFuNetwNML.py:17-141`) and `PIFuMRNet` (fine level,
reference `PIFuMRNet.py:15-243`) keep the reference's method names, argument meaning, cached attributes
(`preds`, `phi`, `intermediate_preds_list`, `im_feat_list`, `nmlF`, `nmlB`, `nmls`, `preds_low`,
`preds_interm`) and call order (filter before query), but hold no torch modules: weights are packed into
a `pifu_ctx` (C ABI, `include/pifuhd_hip.h`) and every query runs in hand-written HIP kernels.
torch tensors are used for device memory and streams only.

Out of scope here (training-time arguments of the same methods): `labels`, `transforms`, perspective
projection, multi-view `B2 > 1`; they raise NotImplementedError instead of silently differing.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_NORM_CODE = {'group': _lib.NORM_GROUP}


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _host_f32(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
    return a, a.ctypes.data_as(C.c_void_p)


class _Ctx:
    """Owns one `pifu_ctx` on one device."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.PifuError(_lib.ERR_INVALID, 'the HIP path needs a GPU device (no CPU fallback)')
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device('cuda', idx)
        h = C.c_void_p()
        _lib.call('pifu_ctx_create', idx, C.byref(h))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                _lib.load().pifu_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass


def _mlp_norm_code(mlp_norm, training):
    if mlp_norm == 'group':
        return _lib.NORM_GROUP
    if mlp_norm == 'batch':
        return _lib.NORM_BATCH_TRAIN if training else _lib.NORM_BATCH_EVAL
    return _lib.NORM_NONE                                  # MLP.py:65-66: anything else = no norm


def _push_mlp(ctx, level, sd, prefix, dims, res_layers, merge_layer, norm_code):
    dims_a = (C.c_int32 * len(dims))(*dims)
    res_a = (C.c_int32 * max(1, len(res_layers)))(*res_layers)
    _lib.call('pifu_mlp_define', ctx.h, level, dims_a, len(dims), res_a, len(res_layers), merge_layer, norm_code)
    n = len(dims) - 1
    keep = []
    for i in range(n):
        w, wp = _host_f32(sd[f'{prefix}.filters.{i}.weight'].reshape(dims[i + 1], -1))
        b, bp = _host_f32(sd[f'{prefix}.filters.{i}.bias'])
        g = be = rm = rv = None
        gp = bep = rmp = rvp = None
        if i != n - 1 and norm_code != _lib.NORM_NONE:
            g, gp = _host_f32(sd[f'{prefix}.norms.{i}.weight'])
            be, bep = _host_f32(sd[f'{prefix}.norms.{i}.bias'])
            if norm_code in (_lib.NORM_BATCH_EVAL, _lib.NORM_BATCH_TRAIN):
                rm, rmp = _host_f32(sd[f'{prefix}.norms.{i}.running_mean'])
                rv, rvp = _host_f32(sd[f'{prefix}.norms.{i}.running_var'])
        keep.append((w, b, g, be, rm, rv))
        _lib.call('pifu_mlp_set_layer', ctx.h, level, i, wp, bp, gp, bep, rmp, rvp)


def _check_state_dict(owner, spec, given_keys, state_dict, strict, prefix=''):
    """The checks of `nn.Module.load_state_dict` (what the reference's `netMR.load_state_dict(...)` runs,
    `reconstruction.py:300`): size mismatches always raise; missing and unexpected keys raise in strict mode, with
    torch's wording."""
    errors = []
    for k, shape in spec.items():
        if k in state_dict and tuple(state_dict[k].shape) != tuple(shape):
            errors.append(f'size mismatch for {k[len(prefix):]}: copying a param with shape {tuple(state_dict[k].shape)} from '
                          f'checkpoint, the shape in current model is {tuple(shape)}.')
    if strict:
        missing = [k[len(prefix):] for k in spec if k not in state_dict]
        unexpected = [k[len(prefix):] for k in given_keys if k not in spec]
        if unexpected:
            errors.insert(0, 'Unexpected key(s) in state_dict: {}. '.format(', '.join(f'"{k}"' for k in unexpected)))
        if missing:
            errors.insert(0, 'Missing key(s) in state_dict: {}. '.format(', '.join(f'"{k}"' for k in missing)))
    if errors:
        raise RuntimeError('Error(s) in loading state_dict for {}:\n\t{}'.format(owner, '\n\t'.join(errors)))


_ENC_NORM = {('batch', False): 0, ('batch', True): 1, ('group', False): 2, ('group', True): 2}


def _push_encoder_params(ctx, sd, prefix):
    """Register every floating-point parameter / buffer of an encoder with the context: convolution weights that live on the
    context's device go across as device pointers (BORROWED by the library: no second copy of the 1.5 GB, no host round trip),
    everything else as HOST fp32 copies.  Returns the tensors the library now points at: the caller keeps them alive until the
    next registration."""
    keep = []
    for key, t in sd.items():
        if not (key.startswith('image_filter.') or key.startswith('netF.') or key.startswith('netB.')):
            continue
        if not t.dtype.is_floating_point:
            continue
        if t.dim() == 4 and t.is_cuda and t.device == ctx.device:
            d = t.detach().to(torch.float32).contiguous()
            shape = (C.c_int64 * 4)(*d.shape)
            _lib.call('pifu_enc_set_param_device', ctx.h, (prefix + key).encode(), _ptr(d), shape, 4, _stream_ptr(ctx.device))
            keep.append(d)
            continue
        a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
        shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
        _lib.call('pifu_enc_set_param', ctx.h, (prefix + key).encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)
    return keep


def orthogonal(points, calib, transform=None):
    """`orthogonal` (`BasePIFuNet.py:25-43`): xyz = R p + t for points [B,3,N], calib [B,4,4] (device tensors)."""
    if transform is not None:
        raise NotImplementedError('screen-space `transforms` are never used on the reconstruction path')
    rot = calib[:, :3, :3]
    trans = calib[:, :3, 3:4]
    return torch.baddbmm(trans, rot, points)


def index(feat, uv, ctx=None):
    """`index` (`BasePIFuNet.py:11-23`): feat [1,C,H,W], uv [1,2,N] -> [1,C,N] through `pifu_index` (HIP)."""
    if feat.dim() != 4 or feat.shape[0] != 1 or uv.dim() != 3 or uv.shape[0] != 1 or uv.shape[1] != 2:
        raise NotImplementedError('index(): batch size 1, feat [1,C,H,W], uv [1,2,N]')
    if not feat.is_cuda:
        raise _lib.PifuError(_lib.ERR_INVALID, 'the HIP path needs device tensors (no CPU fallback)')
    ctx = ctx or _Ctx(feat.device)
    f = feat[0].float().contiguous()
    u = uv[0].to(feat.device, torch.float32).contiguous()
    out = torch.empty(f.shape[0], u.shape[1], device=feat.device, dtype=torch.float32)
    _lib.call('pifu_index', ctx.h, _ptr(f), f.shape[0], f.shape[1], f.shape[2], _ptr(u), u.shape[1], _ptr(out),
              _stream_ptr(ctx.device))
    return out[None]


class BasePIFuNet:
    """API shell (`BasePIFuNet.py:67-148`)."""
    projection = staticmethod(orthogonal)

    def __init__(self, projection_mode='orthogonal'):
        if projection_mode not in ('orthogonal', 'otthogonal'):      # 'otthogonal': PIFuMRNet.py:22 default typo
            raise NotImplementedError('only the orthogonal projection is on the reconstruction path '
                                      '(readData.py:34); perspective is out of scope')
        self.name = 'base'
        self.preds = None
        self.labels = None
        self.nmls = None
        self.training = True

    def forward(self, points, images, calibs, transforms=None):
        self.filter(images)
        self.query(points, calibs, transforms)
        return self.get_preds()

    __call__ = forward

    def get_preds(self):
        return self.preds

    def calc_normal(self, points, calibs, transforms=None, delta=0.1):
        """`BasePIFuNet.calc_normal` (`BasePIFuNet.py:125-134`): a placeholder in the reference (its body is `None`)."""
        return None

    def get_error(self, gamma=None):
        """`BasePIFuNet.get_error` (`BasePIFuNet.py:144-148`) calls `self.error_term`, which no class of the reference
        defines: the call raises AttributeError there, and here."""
        return self.error_term(self.preds, self.labels, gamma)

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    @staticmethod
    def _check_unsupported(transforms, labels=None):
        if transforms is not None:
            raise NotImplementedError('screen-space `transforms` are never used on the reconstruction path')


class PIFuNetwNML(BasePIFuNet):
    """Coarse level: normal nets + HGFilter + SurfaceClassifier (`PIFuNetwNML.py:17-141`)."""

    def __init__(self, opt, projection_mode='orthogonal', device='cuda', criteria=None):
        super().__init__(projection_mode)
        self.name = 'hg_pifu'
        self.opt = opt
        # the reference's default {'occ': nn.MSELoss()} cannot take get_error's (pred, gt, gamma) call; its training script
        # passes CustomBCELoss, which is the default here (as in PIFuMRNet)
        from .net_util import CustomBCELoss
        self.criteria = criteria if criteria is not None else {'occ': CustomBCELoss()}
        self.nml = None                  # what calc_normal leaves behind (the reference's attribute name, SURVEY C-9)
        self.labels_nml = None
        self._ctx_obj = _Ctx(device)
        self.device = self._ctx_obj.device
        self._sd = None
        self.im_feat_list = []
        self.normx = None
        self.phi = None
        self.intermediate_preds_list = []
        self.nmlF = None
        self.nmlB = None
        self.netF = 'netF' if getattr(opt, 'use_front_normal', False) else None
        self.netB = 'netB' if getattr(opt, 'use_back_normal', False) else None
        self._last_n = 0
        self._dirty, self._pushed_code = True, None
        self._enc_dirty, self._enc_keep = True, []      # the encoder tensors changed since they were registered (load_state_dict does)

    @property
    def _ctx(self):
        """The library context, with pending weight changes (load_state_dict, train / eval) uploaded first: whoever calls
        the C ABI through it sees the current weights."""
        if self._sd is not None:
            self._sync_weights()
        return self._ctx_obj

    # -- parameters --------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True, _prefix='', _check=True):
        from . import synth
        spec = synth.coarse_spec(self.opt, _prefix)
        if _check:
            _check_state_dict('PIFuNetwNML', spec, [k for k in state_dict if k.startswith(_prefix)], state_dict, strict, _prefix)
        self._sd = {k[len(_prefix):]: state_dict[k].detach().to(self.device, torch.float32)
                    if state_dict[k].dtype.is_floating_point else state_dict[k]
                    for k in spec if k in state_dict}
        self._dirty = self._enc_dirty = True

    def state_dict(self):
        """Flat dict keyed like the reference module's `state_dict()` (no 'netG.' prefix)."""
        if self._sd is None:
            raise _lib.PifuError(_lib.ERR_STATE, 'no weights loaded')
        return dict(self._sd)

    def loadFromPIFu(self, net):
        """`PIFuNetwNML.loadFromPIFu` (`PIFuNetwNML.py:144-179`): copy every `image_filter.*` / `mlp.*` tensor of `net`
        whose key and shape match this net's; prints the top-level names left uninitialised, like the reference.
        (The reference starts from its random initialisation; here this net must already hold weights.)"""
        mine, theirs = self.state_dict(), net.state_dict()
        for part, label in (('image_filter.', 'not initialized'), ('mlp.', 'mlp not initialized')):
            not_init = set()
            for k, v in mine.items():
                if not k.startswith(part):
                    continue
                t = theirs.get(k)
                if t is not None and tuple(t.shape) == tuple(v.shape):
                    mine[k] = t.detach().to(self.device, v.dtype)
                else:
                    not_init.add(k[len(part):].split('.')[0])
            print(label, sorted(not_init))
        self._sd = mine
        self._dirty = self._enc_dirty = True

    def _push_weights(self):
        o = self.opt
        # eval mode on the reconstruction path (reconstruction.py:288-289): running stats for 'batch'
        code = _mlp_norm_code(o.mlp_norm, self.training)
        ctx = self._ctx_obj
        _push_mlp(ctx, _lib.LEVEL_COARSE, self._sd, 'mlp', list(o.mlp_dim), list(o.mlp_res_layers), int(o.merge_layer), code)
        _lib.call('pifu_set_depth_normalizer', ctx.h, float(o.loadSize // 2), float(o.z_size))
        # the encoders' tensors go across only when they changed (an explicit load): a re-push for the MLPs' sake - train / eval,
        # a bias shifted by synth.calibrate_occupancy_bias - must not make the library drop and repack 1.5 GB of convolution weights
        if self._enc_dirty:
            self._enc_keep = _push_encoder_params(ctx, self._sd, '')      # borrowed by the library until the next push
            self._enc_dirty = False
        self._dirty, self._pushed_code = False, code

    def _sync_weights(self):
        """Upload the weights when they (or the train/eval mode the MLP norm depends on) changed since the last upload:
        `load_state_dict` and `train()` only mark, so `recon()`'s load -> `eval()` sequence uploads once."""
        if self._sd is None:
            raise _lib.PifuError(_lib.ERR_STATE, 'load_state_dict() must precede filter() / query()')
        if self._dirty or self._pushed_code != _mlp_norm_code(self.opt.mlp_norm, self.training):
            self._push_weights()

    def train(self, mode=True):
        self.training = mode
        return self

    # -- filter ------------------------------------------------------------------------------------
    def filter(self, images, _defer=False, _part=None):
        """`PIFuNetwNML.filter` (`PIFuNetwNML.py:73-97`).  `_defer=True` (used by `reconstruction._Subject`, which calls
        `filter_local` next): return while the hourglass is still running on a stream of the library, so that the fine
        level runs beside it; `im_feat_list` must then not be read before `filter_local` / a query has been issued.
        `_part` (`dist.filter_sharded`: the call dealt out over ranks): 'F' / 'B' = that normal net alone (fills
        `self.nmlF` / `self.nmlB`), 'hourglass' = the HGFilter on the normal maps `self.nmlF` / `self.nmlB` hold."""
        self._sync_weights()
        images = images.to(self.device, torch.float32).contiguous()
        if images.dim() != 4 or images.shape[0] != 1:
            raise NotImplementedError('images [1,C,H,W] (batch size 1, the reconstruction path)')
        # with _defer the hourglass still reads `images` on a stream of the library after this method returned; torch's
        # allocator does not know that stream, so a converted temporary must not go back to it before the join: the net holds
        # the tensor until the next filter() call replaces it
        self._coarse_in = images
        o = self.opt
        _, cimg, h, w = images.shape
        if o.hg_down != 'ave_pool':
            if o.hg_down == 'no_down':
                raise NotImplementedError("coarse level with hg_down='no_down'")
            raise NameError('unknown downsampling type')                       # Filter.py:197
        use_f, use_b = self.netF is not None, self.netB is not None
        flags = (_lib.ENC_ALL_STACKS if self.training else 0) | (_lib.ENC_DEFER_JOIN if _defer else 0)
        if getattr(self, '_direct_trunk', False):           # A/B hook of the tests: the normal nets' trunks as direct convolutions
            flags |= _lib.ENC_NO_WINOGRAD
        run_f, run_b = use_f, use_b                      # which normal nets the call enables
        if _part in ('F', 'B'):
            run_f, run_b = use_f and _part == 'F', use_b and _part == 'B'
            flags |= _lib.ENC_NORMALS_ONLY
        elif _part == 'hourglass':
            flags |= _lib.ENC_NORMALS_GIVEN
        elif _part is not None:
            raise ValueError("_part must be None, 'F', 'B' or 'hourglass'")

        def normal_map(old, wanted, fresh):
            if not wanted:
                return None
            if fresh or old is None or tuple(old.shape) != (1, 3, h, w):
                return torch.empty(1, 3, h, w, device=self.device)
            return old                                   # a part of a dealt-out call: the other parts' maps stay
        self.nmlF = normal_map(getattr(self, 'nmlF', None), use_f, _part is None or _part == 'F')
        self.nmlB = normal_map(getattr(self, 'nmlB', None), use_b, _part is None or _part == 'B')
        # train mode (train_full_pifu): BatchNorm2d of the HGFilter on batch statistics and EVERY stack's output kept
        # (PIFuNetwNML.py:94-97); eval mode: running statistics, last stack only
        stacks = int(o.num_stack) if self.training else 1
        feat = torch.empty(stacks, o.hg_dim, h // 4, w // 4, device=self.device) if _part in (None, 'hourglass') else None
        _lib.call('pifu_filter_coarse', self._ctx.h, _ptr(images), cimg, h, w, int(o.num_stack), int(o.hg_depth),
                  int(o.hg_dim), _ENC_NORM[(o.norm, bool(self.training))], 1, int(run_f), int(run_b),
                  _ptr(self.nmlF), _ptr(self.nmlB), _ptr(feat), flags, _stream_ptr(self.device))
        if feat is not None:
            self.im_feat_list = [feat[k:k + 1] for k in range(stacks)]
        self.normx = None                   # never read on the reconstruction path

    def _set_features(self, feat):
        feat = feat.contiguous()
        _, c, h, w = feat.shape
        _lib.call('pifu_set_features', self._ctx.h, _lib.LEVEL_COARSE, _ptr(feat), c, h, w, _stream_ptr(self.device))
        self._feat_keepalive = feat

    def set_features(self, feat):
        """Install a precomputed coarse feature map [1,C,H,W] (what filter() would cache)."""
        feat = feat.to(self.device, torch.float32)
        self.im_feat_list = [feat]
        self._set_features(feat)

    # -- query -------------------------------------------------------------------------------------
    def query(self, points, calibs, transforms=None, labels=None, update_pred=True, update_phi=True):
        """`PIFuNetwNML.query` (`PIFuNetwNML.py:99-141`): points [1,3,N], calibs [1,4,4]."""
        self._check_unsupported(transforms, labels)
        if not self.im_feat_list:
            raise _lib.PifuError(_lib.ERR_STATE, 'filter() must precede query()')
        self._sync_weights()
        pts = points.to(self.device, torch.float32).contiguous()
        if pts.dim() != 3 or pts.shape[0] != 1 or pts.shape[1] != 3:
            raise NotImplementedError('batch size 1, points [1,3,N] (the reconstruction path)')
        n = pts.shape[2]
        calib, calib_p = _host_f32(calibs[0])
        cphi = self.opt.mlp_dim[(self.opt.merge_layer if self.opt.merge_layer > 0 else len(self.opt.mlp_dim) // 2) + 1]
        phi = torch.empty(cphi, n, device=self.device, dtype=torch.float32) if update_phi else None
        # one pass of the (shared) MLP per kept stack (PIFuNetwNML.py:124-134): all of them in train mode, the last one
        # otherwise; phi is the last stack's.  The context ends up holding the LAST stack's map, which is what the fine
        # level's own coarse pass must see.
        preds = []
        for k, feat in enumerate(self.im_feat_list):
            last = k == len(self.im_feat_list) - 1
            if len(self.im_feat_list) > 1:
                self._set_features(feat)
            pred = torch.empty(n, device=self.device, dtype=torch.float32)
            _lib.call('pifu_query', self._ctx.h, _ptr(pts), n, calib_p, n, _lib.Q_COARSE_ONLY, None, _ptr(pred),
                      _ptr(phi) if last else None, _stream_ptr(self.device))
            preds.append(pred[None, None])
        if labels is not None:                                  # PIFuNetwNML.py:115-120: in-box over x, y and z
            xyz = self.projection(pts, calibs.to(self.device, torch.float32))
            in_bb = ((xyz >= -1) & (xyz <= 1)).all(dim=1, keepdim=True).float()
            self.labels = in_bb * labels.to(self.device, torch.float32)
        if update_phi:
            self.phi = phi[None]
        if update_pred:
            self.preds = preds[-1]
            self.intermediate_preds_list = preds


    def calc_normal(self, points, calibs, transforms=None, labels=None, delta=0.1, keep_pred=False):
        """`PIFuNetwNML.calc_normal` (`PIFuNetwNML.py:181-218`): points [1,3,N], calibs [1,4,4] -> `self.nml` [1,3,N], the
        negated, normalised forward differences (step `delta`, 0.1 by default) of the coarse net's own UNMASKED occupancy;
        the 4N points [p, p+dx, p+dy, p+dz] go through the MLP as ONE call (one GroupNorm batch), as in the reference.
        Like the reference it writes `self.nml` - `get_error` reads `self.nmls`, which this method never sets (SURVEY C-9)."""
        self._check_unsupported(transforms)
        if not self.im_feat_list:
            raise _lib.PifuError(_lib.ERR_STATE, 'filter() must precede calc_normal()')
        if labels is not None:
            self.labels_nml = labels
        self._sync_weights()
        pts = points.to(self.device, torch.float32)
        if pts.dim() != 3 or pts.shape[0] != 1 or pts.shape[1] != 3:
            raise NotImplementedError('batch size 1, points [1,3,N] (the reconstruction path)')
        pts = pts[0].contiguous()
        n = pts.shape[1]
        if n == 0:
            raise RuntimeError('calc_normal on an empty point list')      # Conv1d on an empty tensor raises in the reference
        if len(self.im_feat_list) > 1:
            self._set_features(self.im_feat_list[-1])                      # PIFuNetwNML.py:203: the LAST stack's map
        calib, calib_p = _host_f32(calibs[0])
        out = torch.empty(3, n, device=self.device, dtype=torch.float32)
        pred4 = torch.empty(4 * n, device=self.device, dtype=torch.float32) if keep_pred else None
        _lib.call('pifu_calc_normal_coarse', self._ctx.h, _ptr(pts), n, calib_p, float(delta), _ptr(out), _ptr(pred4),
                  _stream_ptr(self.device))
        self.nml = out[None]
        self.nml_preds = pred4.view(1, 1, n, 4) if keep_pred else None     # `pred` of PIFuNetwNML.py:209, [B, 1, N, 4]

    def get_im_feat(self):
        """`PIFuNetwNML.get_im_feat` (`PIFuNetwNML.py:220-226`)."""
        return self.im_feat_list[-1]

    def get_error(self, gamma):
        """`PIFuNetwNML.get_error` (`PIFuNetwNML.py:228-243`), forward value only: the occupancy criterion averaged over the
        stacks' predictions with the scalar `gamma`; `Err(nml)` only when `self.nmls` and the normal labels both exist."""
        if self.labels is None or not self.intermediate_preds_list:
            raise _lib.PifuError(_lib.ERR_STATE, 'query(..., labels=...) must precede get_error()')
        g = torch.tensor([float(gamma)])
        total = 0.0
        for preds in self.intermediate_preds_list:
            total = total + self.criteria['occ'](preds, self.labels, g)
        error = {'Err(occ)': total / len(self.intermediate_preds_list)}
        if self.nmls is not None and self.labels_nml is not None:
            error['Err(nml)'] = self.criteria['nml'](self.nmls, self.labels_nml)
        return error

    def forward(self, images, points, calibs, labels, gamma, points_nml=None, labels_nml=None):
        """`PIFuNetwNML.forward` (`PIFuNetwNML.py:245-258`), forward-only (no autograd on the HIP path) -> (err, res)."""
        self.filter(images)
        self.query(points, calibs, labels=labels)
        if points_nml is not None and labels_nml is not None:
            self.calc_normal(points_nml, calibs, labels=labels_nml)
        return self.get_error(gamma), self.get_preds()

    __call__ = forward


class PIFuMRNet(BasePIFuNet):
    """Fine level (`PIFuMRNet.py:15-243`); owns a reference to the coarse net `netG`."""

    def __init__(self, opt, netG, projection_mode='orthogonal', criteria=None):
        super().__init__(projection_mode)
        self.name = 'hg_pifu'
        self.opt = opt
        self.netG = netG
        # the reference's default {'occ': nn.MSELoss()} cannot take get_error's (pred, gt, gamma, w) call; its training
        # script passes CustomBCELoss (train.py), which is the default here
        from .net_util import CustomBCELoss
        self.criteria = criteria if criteria is not None else {'occ': CustomBCELoss()}
        self.w = None
        self.gamma = None
        self.labels_nml = None
        self.device = netG.device
        self._sd = None
        self.im_feat_list = []
        self.preds_interm = None
        self.preds_low = None
        self.normx = None
        self.training = True          # a freshly constructed module is in train mode (SURVEY C-2)
        self._dirty, self._pushed_code = True, None
        self._enc_dirty, self._enc_keep = True, []

    @property
    def _ctx(self):
        """The context shared with netG, with the pending weight changes of both levels uploaded first."""
        if self._sd is not None:
            self._sync_weights()
        return self.netG._ctx_obj

    def train(self, mode=True):
        """`PIFuMRNet.train` (`PIFuMRNet.py:59-69`): netG is forced to eval unless train_full_pifu."""
        self.training = mode
        if not getattr(self.opt, 'train_full_pifu', False):
            self.netG.eval()
        return self

    def load_state_dict(self, state_dict, strict=True):
        from . import synth
        _check_state_dict('PIFuMRNet', synth.state_dict_spec(self.netG.opt, self.opt), list(state_dict), state_dict, strict)
        self.netG.load_state_dict(state_dict, strict, _prefix='netG.', _check=False)
        own = {}
        synth.filter_keys('image_filter', self.opt.num_stack, self.opt.hg_depth,
                          synth.image_filter_in_channels(self.netG.opt), self.opt.hg_dim, self.opt.norm, 'no_down', own)
        synth.mlp_keys('mlp', self.opt.mlp_dim, self.opt.mlp_res_layers, self.opt.mlp_norm, own)
        self._sd = {k: state_dict[k].detach().to(self.device, torch.float32)
                    if state_dict[k].dtype.is_floating_point else state_dict[k] for k in own if k in state_dict}
        self._dirty = self._enc_dirty = True

    def state_dict(self):
        """Flat dict with the reference's key scheme: own tensors + the coarse net's under 'netG.'."""
        if self._sd is None:
            raise _lib.PifuError(_lib.ERR_STATE, 'no weights loaded')
        out = dict(self._sd)
        out.update({'netG.' + k: v for k, v in self.netG.state_dict().items()})
        return out

    def _push_weights(self):
        o = self.opt
        code = _mlp_norm_code(o.mlp_norm, self.training)      # train mode => per-chunk batch statistics (C-2)
        self.netG._sync_weights()                               # the fine MLP is defined against the coarse one
        ctx = self.netG._ctx_obj
        _push_mlp(ctx, _lib.LEVEL_FINE, self._sd, 'mlp', list(o.mlp_dim), list(o.mlp_res_layers), -1, code)
        if self._enc_dirty:
            self._enc_keep = _push_encoder_params(ctx, self._sd, 'fine.')
            self._enc_dirty = False
        self._dirty, self._pushed_code = False, code

    def _sync_weights(self):
        if self._sd is None:
            raise _lib.PifuError(_lib.ERR_STATE, 'load_state_dict() must precede filter_local() / query()')
        coarse_repush = self.netG._dirty or self.netG._pushed_code != _mlp_norm_code(self.netG.opt.mlp_norm, self.netG.training)
        self.netG._sync_weights()
        # re-defining the coarse MLP invalidates the fine definition (pifu_mlp_define of the coarse level resets both)
        if coarse_repush or self._dirty or self._pushed_code != _mlp_norm_code(self.opt.mlp_norm, self.training):
            self._push_weights()

    # -- filter ------------------------------------------------------------------------------------
    def filter_global(self, images, _defer=False):
        """`PIFuMRNet.filter_global` (`PIFuMRNet.py:71-81`); `_defer`: see `PIFuNetwNML.filter`."""
        self.netG.filter(images, _defer=_defer)

    def filter_local(self, images, rect=None):
        """`PIFuMRNet.filter_local` (`PIFuMRNet.py:83-118`): images [1,1,C,H,W]."""
        if rect is not None:
            raise NotImplementedError('`rect` crops are never used on the reconstruction path')
        self._sync_weights()
        images = images.to(self.device, torch.float32)
        if images.dim() != 5 or images.shape[0] != 1 or images.shape[1] != 1:
            raise NotImplementedError('images [1,1,C,H,W] (single view, the reconstruction path)')
        x = images[0].contiguous()
        _, cimg, h, w = x.shape
        o, g = self.opt, self.netG
        use_f = bool(getattr(g.opt, 'use_front_normal', False))
        use_b = bool(getattr(g.opt, 'use_back_normal', False))
        if (use_f and g.nmlF is None) or (use_b and g.nmlB is None):
            raise _lib.PifuError(_lib.ERR_STATE, 'filter_global() must precede filter_local()')
        big = int(o.loadSizeBig) if (use_f or use_b) else h
        if (use_f or use_b) and (h != big or w != big):
            raise ValueError(f'image size {h}x{w} does not match loadSizeBig {big} (torch.cat would fail in the reference)')
        hn, wn = (g.nmlF if use_f else g.nmlB).shape[2:] if (use_f or use_b) else (h, w)
        feat = torch.empty(1, o.hg_dim, h // 2, w // 2, device=self.device)
        _lib.call('pifu_filter_fine', self._ctx.h, _ptr(x), cimg, h, w, _ptr(g.nmlF) if use_f else None,
                  _ptr(g.nmlB) if use_b else None, int(hn), int(wn), int(o.num_stack), int(o.hg_depth), int(o.hg_dim),
                  _ENC_NORM[(o.norm, bool(self.training))], _ptr(feat), _stream_ptr(self.device))
        if self.training and o.num_stack > 1:
            raise NotImplementedError('train-mode multi-stack outputs are a training-time path')
        self.im_feat_list = [feat]
        self.normx = None

    def _set_features(self, feat):
        feat = feat.contiguous()
        _, c, h, w = feat.shape
        _lib.call('pifu_set_features', self._ctx.h, _lib.LEVEL_FINE, _ptr(feat), c, h, w, _stream_ptr(self.device))
        self._feat_keepalive = feat

    def set_features(self, feat):
        feat = feat.to(self.device, torch.float32)
        self.im_feat_list = [feat]
        self._set_features(feat)

    # -- query -------------------------------------------------------------------------------------
    def query(self, points, calib_local, calib_global=None, transforms=None, labels=None):
        """`PIFuMRNet.query` (`PIFuMRNet.py:119-186`), 3-argument form used by reconstruction:
        points [1,3,N], calib_local [1,4,4]."""
        self._check_unsupported(transforms, labels)
        if calib_global is not None:
            if calib_local.shape[1] != 1:
                raise NotImplementedError('multi-view B2 > 1 is out of scope')
            points = points[:, 0]
            calib_local = calib_local[:, 0]
            if not torch.equal(calib_local.cpu(), calib_global.cpu()):
                raise NotImplementedError('different local/global calibrations (crop training) are out of scope')
        if not self.im_feat_list or not self.netG.im_feat_list:
            raise _lib.PifuError(_lib.ERR_STATE, 'filter_global()/filter_local() must precede query()')
        self._sync_weights()
        pts = points.to(self.device, torch.float32).contiguous()
        if pts.dim() != 3 or pts.shape[0] != 1 or pts.shape[1] != 3:
            raise NotImplementedError('batch size 1, points [1,3,N] (the reconstruction path)')
        n = pts.shape[2]
        calib, calib_p = _host_f32(calib_local[0])
        fine = torch.empty(n, device=self.device, dtype=torch.float32)
        coarse = torch.empty(n, device=self.device, dtype=torch.float32)
        g = self.netG
        cphi = g.opt.mlp_dim[(g.opt.merge_layer if g.opt.merge_layer > 0 else len(g.opt.mlp_dim) // 2) + 1]
        phi = torch.empty(cphi, n, device=self.device, dtype=torch.float32)
        if len(g.im_feat_list) > 1:
            # train_full_pifu: netG kept every stack; its query runs the MLP on each of them (PIFuMRNet.py:154-155) and
            # leaves the last stack's map installed, which the fused coarse + fine pass below then reads
            g.query(pts, calib_local)
            earlier = g.intermediate_preds_list[:-1]
        else:
            earlier = []
        _lib.call('pifu_query', self._ctx.h, _ptr(pts), n, calib_p, n, 0, _ptr(fine), _ptr(coarse), _ptr(phi),
                  _stream_ptr(self.device))
        # attributes the reference leaves behind (PIFuMRNet.py:154-181, PIFuNetwNML.py:137-141)
        g.phi = phi[None]
        g.preds = coarse[None, None]
        g.intermediate_preds_list = earlier + [g.preds]
        self.preds = fine[None, None]
        self.preds_interm = self.preds[None]
        self.preds_low = torch.stack(g.intermediate_preds_list, 0)
        if labels is not None:                                  # PIFuMRNet.py:157-162,183-186
            lab = labels.to(self.device, torch.float32)
            if lab.dim() == 4:
                lab = lab[:, 0]
            xyz = self.projection(pts, calib_local.to(self.device, torch.float32))
            in_bb = ((xyz[:, :2] >= -1) & (xyz[:, :2] <= 1)).all(dim=1, keepdim=True).float()
            self.labels = in_bb * lab
            self.w = in_bb.size(2) / in_bb.view(in_bb.size(0), -1).sum(1)
            self.gamma = 1 - self.labels.view(self.labels.size(0), -1).sum(1) / in_bb.view(in_bb.size(0), -1).sum(1)

    def get_error(self):
        """`PIFuMRNet.get_error` (`PIFuMRNet.py:253-274`): the occupancy criterion averaged over a stack of predictions.
        `Err(occ)` (only with train_full_pifu and intermediate losses on) runs over the coarse stacks `preds_low`;
        `Err(occ:fine)` runs over as many entries as `preds_interm` has but - like the reference - READS `preds_low`;
        `Err(nml:fine)` appears when normals and their labels are both present."""
        occ = self.criteria['occ']

        def averaged(count):
            total = 0.0
            for i in range(count):
                total = total + occ(self.preds_low[i], self.labels, self.gamma, self.w)
            return total / count
        error = {}
        if getattr(self.opt, 'train_full_pifu', False) and not getattr(self.opt, 'no_intermediate_loss', False):
            error['Err(occ)'] = averaged(self.preds_low.size(0))
        error['Err(occ:fine)'] = averaged(self.preds_interm.size(0))
        if self.nmls is not None and self.labels_nml is not None:
            error['Err(nml:fine)'] = self.criteria['nml'](self.nmls, self.labels_nml)
        return error

    def forward(self, images_local, images_global, points, calib_local, calib_global, labels, points_nml=None,
                labels_nml=None, rect=None):
        """`PIFuMRNet.forward` (`PIFuMRNet.py:276-288`): forward-only (no autograd on the HIP path) -> (err, res)."""
        self.filter_global(images_global)
        self.filter_local(images_local, rect)
        self.query(points, calib_local, calib_global, labels=labels)
        if points_nml is not None and labels_nml is not None:
            self.calc_normal(points_nml, calib_local, calib_global, labels=labels_nml)
        return self.get_error(), self.get_preds()

    __call__ = forward

    def calc_normal(self, points, calib_local, calib_global, transforms=None, labels=None, delta=0.001,
                    fd_type='forward', keep_pred=False):
        """`PIFuMRNet.calc_normal` (`PIFuMRNet.py:188-243`): points [1,1,3,n] -> self.nmls [1,3,n]."""
        self._check_unsupported(transforms)
        if labels is not None:
            self.labels_nml = labels
        pts = points.to(self.device, torch.float32)
        if pts.dim() != 4 or pts.shape[0] != 1 or pts.shape[1] != 1 or pts.shape[2] != 3:
            raise NotImplementedError('points [1,1,3,n] (single view, the reconstruction path)')
        pts = pts[0, 0].contiguous()
        n = pts.shape[1]
        if n == 0:
            # the reference crashes inside Conv1d on an empty slice (SURVEY C-6)
            raise RuntimeError('calc_normal on an empty vertex slice')
        self._sync_weights()
        calib, calib_p = _host_f32(calib_global[0])
        out = torch.empty(3, n, device=self.device, dtype=torch.float32)
        pred4 = torch.empty(4 * n, device=self.device, dtype=torch.float32) if keep_pred else None
        _lib.call('pifu_calc_normal', self._ctx.h, _ptr(pts), n, calib_p, float(delta), _ptr(out), _ptr(pred4),
                  _stream_ptr(self.device))
        self.nmls = out[None]
        self.nml_preds = pred4.view(1, 1, n, 4) if keep_pred else None      # `pred` of PIFuMRNet.py:225, [B1, 1, N, 4]

    def calc_normal_chunked(self, points, calib_local, calib_global, interval=50000, delta=0.001):
        """The colour loop of `gen_mesh` (`reconstruction.py:58-70`) in one call: the same numbers as `calc_normal` on
        consecutive `interval`-vertex slices of points [1,1,3,n] (each slice its own GroupNorm batch) -> self.nmls [1,3,n]."""
        pts = points.to(self.device, torch.float32)
        if pts.dim() != 4 or pts.shape[0] != 1 or pts.shape[1] != 1 or pts.shape[2] != 3:
            raise NotImplementedError('points [1,1,3,n] (single view, the reconstruction path)')
        pts = pts[0, 0].contiguous()
        n = pts.shape[1]
        if n == 0:
            raise RuntimeError('calc_normal on an empty vertex slice')
        self._sync_weights()
        calib, calib_p = _host_f32(calib_global[0])
        out = torch.empty(3, n, device=self.device, dtype=torch.float32)
        _lib.call('pifu_calc_normal_chunked', self._ctx.h, _ptr(pts), n, int(interval), calib_p, float(delta), _ptr(out),
                  None, _stream_ptr(self.device))
        self.nmls = out[None]

    def get_im_feat(self):
        return self.im_feat_list[-1]


def build_nets(opt_coarse, opt_fine, state_dict, device='cuda'):
    """What `recon()` does before its loop (`reconstruction.py:283-304`): build both nets, load one state
    dict, put ONLY netG in eval mode."""
    netG = PIFuNetwNML(opt_coarse, 'orthogonal', device)
    netMR = PIFuMRNet(opt_fine, netG, 'orthogonal')
    netG.eval()
    netMR.load_state_dict(state_dict)
    return netG, netMR
