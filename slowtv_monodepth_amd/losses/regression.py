"""`RegressionLoss` — registry keys `depth_regr`, `stereo_const` (reference: `src/losses/regression.py:40-75`)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..registry import register

__all__ = ['RegressionLoss']


@register(('depth_regr', 'stereo_const'))
class RegressionLoss(nn.Module):
    """Supervised / consistency regression loss: masked mean of a dense error, evaluated by `smd_regression_*`.

    :param loss_name: 'l1', 'log_l1' (Depth Hints) or 'berhu' (Kuznietsov; dynamic threshold 0.2*max|pred - target|).
    :param invert: convert both inputs depth -> disparity (`to_inv`) first.
    :param use_automask: read by `handlers.depth_regr`, which builds the Depth-Hints automask and passes it in as `mask`.
    """
    def __init__(self, loss_name: str = 'berhu', invert: bool = False, use_automask: bool = False):
        super().__init__()
        if loss_name not in {'l1', 'log_l1', 'berhu'}: raise KeyError(loss_name)
        self.loss_name, self.invert, self.use_automask = loss_name, invert, use_automask

    def forward(self, pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor | None = None):
        """:return: (loss (), {'err_regr': masked dense error, 'mask_regr': the mask used (all ones if None)})"""
        from .. import functional as F
        loss, err = F.regression_loss(pred, target, mask, loss_name=self.loss_name, invert=self.invert)
        return loss, {'err_regr': err, 'mask_regr': mask if mask is not None else torch.ones_like(target)}
