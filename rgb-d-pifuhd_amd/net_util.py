"""Losses of the reference's `net_util.py` used by the supervised forward (`PIFuMRNet.get_error`, `PIFuMRNet.py:253-274`).
Element-wise torch ops on the device tensors the HIP query produced: a few bytes per point, not on the hot path."""
from __future__ import annotations

import torch


class CustomBCELoss:
    """`CustomBCELoss` (`net_util.py:45-66`)."""

    def __init__(self, brock=False, gamma=None):
        self.brock = brock
        self.gamma = gamma

    def forward(self, pred, gt, gamma, w=None):
        x_hat = torch.clamp(pred, 1e-5, 1.0 - 1e-5)
        gamma = gamma[:, None, None] if self.gamma is None else self.gamma
        if self.brock:
            x = 3.0 * gt - 1.0
            loss = -(gamma * x * torch.log(x_hat) + (1.0 - gamma) * (1.0 - x) * torch.log(1.0 - x_hat))
        else:
            loss = -(gamma * gt * torch.log(x_hat) + (1.0 - gamma) * (1.0 - gt) * torch.log(1.0 - x_hat))
        if w is not None:
            if len(w.size()) == 1:
                w = w[:, None, None]
            return (loss * w).mean()
        return loss.mean()

    __call__ = forward
