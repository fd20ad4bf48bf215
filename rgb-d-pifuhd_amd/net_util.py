"""Loss of the supervised forward (`PIFuMRNet.get_error`, reference `PIFuMRNet.py:253-274`): the weighted, class-balanced
binary cross entropy the reference's training script installs as `criteria['occ']` (`net_util.py:45-66`).

Forward value only - there is no autograd on the HIP path.  The reduction runs in `pifu_bce_loss` (csrc/loss.hip): one
pass over the predictions with fixed-order double accumulation, so the value is reproducible run to run.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class CustomBCELoss:
    """Callable with the reference's signature `(pred, gt, gamma, w=None)`.

    pred / gt: device tensors of one shape whose first axis is the batch (`[B, 1, N]` in `get_error`); gamma: `[B]`
    per-sample positive-class weight, ignored when the loss object was built with its own `gamma`; w: `None`, `[B]`
    per-sample weights, or a tensor of pred's shape.  `brock=True` rescales the labels to [-1, 2] (3 gt - 1)."""

    def __init__(self, brock=False, gamma=None):
        self.brock = bool(brock)
        self.gamma = gamma

    def __call__(self, pred, gt, gamma, w=None):
        from .nets import _Ctx, _ptr, _stream_ptr
        if not pred.is_cuda:
            raise _lib.PifuError(_lib.ERR_INVALID, 'the HIP path needs device tensors (no CPU fallback)')
        dev = pred.device
        p = pred.detach().to(torch.float32).contiguous()
        t = gt.detach().to(dev, torch.float32).expand_as(p).contiguous()
        batch = p.shape[0]
        n = p.numel() // batch
        g_ptr, fixed = None, 0.0
        if self.gamma is None:
            g = gamma.detach().to(dev, torch.float32).reshape(-1).contiguous()
            if g.numel() != batch:
                raise ValueError(f'gamma must hold one value per batch entry ({batch}), got {g.numel()}')
            g_ptr = _ptr(g)
        else:
            fixed = float(self.gamma)
        w_ptr, per_element = None, 0
        if w is not None:
            wt = w.detach().to(dev, torch.float32)
            if wt.dim() == 1:
                if wt.numel() != batch:
                    raise ValueError(f'w must hold one value per batch entry ({batch}), got {wt.numel()}')
                wt = wt.contiguous()
            else:
                wt, per_element = wt.expand_as(p).contiguous(), 1
            w_ptr = _ptr(wt)
        ctx = _ctx_cache(dev)
        out = torch.empty(1, device=dev, dtype=torch.float32)
        _lib.call('pifu_bce_loss', ctx.h, _ptr(p), _ptr(t), batch, n, g_ptr, C.c_float(fixed), w_ptr, per_element,
                  int(self.brock), _ptr(out), _stream_ptr(dev))
        return out[0]

    forward = __call__


_CTXS = {}


def _ctx_cache(device):
    """One library context per device for the stand-alone ops of this module."""
    from .nets import _Ctx
    key = torch.device(device).index or 0
    if key not in _CTXS:
        _CTXS[key] = _Ctx(torch.device('cuda', key))
    return _CTXS[key]
