"""Loss modules of the GGA head config.

The reference takes these two from the un-vendored ``mmdet`` wheel
(``mmdet.models.losses``; pinned 2.24-3.0 by mmdet3d/__init__.py:31-32) and
builds them from ``configs/gga/gga_kitti_config.py:59-60``:

    loss_cls  = GaussianFocalLoss(reduction='mean', alpha=0.)
    loss_bbox = L1Loss(reduction='mean', loss_weight=0.25)

Their arithmetic is restated here from mmdet's published algorithm (parity
unpinned: no reference test covers them):

* gaussian focal loss: ``-log(p+1e-12)(1-p)^a [t==1] - log(1-p+1e-12) p^a (1-t)^g``
* weighted mean with ``avg_factor``: ``sum(loss*w) / (avg_factor + eps_f32)``

These eager modules are the *module-level API* (what ``build_loss`` returns);
the fused HIP path in ``gga_amd.functional`` computes the same numbers.
"""
import torch
from torch import nn

from .registry import LOSSES

_EPS32 = torch.finfo(torch.float32).eps


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / (avg_factor + _EPS32)
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def gaussian_focal_loss(pred, gaussian_target, alpha=2.0, gamma=4.0):
    eps = 1e-12
    pos_weights = gaussian_target.eq(1)
    neg_weights = (1 - gaussian_target).pow(gamma)
    pos_loss = -(pred + eps).log() * (1 - pred).pow(alpha) * pos_weights
    neg_loss = -(1 - pred + eps).log() * pred.pow(alpha) * neg_weights
    return pos_loss + neg_loss


@LOSSES.register_module()
class GaussianFocalLoss(nn.Module):
    def __init__(self, alpha=2.0, gamma=4.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None,
                reduction_override=None):
        reduction = reduction_override or self.reduction
        loss = gaussian_focal_loss(pred, target, self.alpha, self.gamma)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction,
                                                     avg_factor)


@LOSSES.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None,
                reduction_override=None):
        reduction = reduction_override or self.reduction
        if target.numel() == 0:
            return pred.sum() * 0
        loss = torch.abs(pred - target)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction,
                                                     avg_factor)


@LOSSES.register_module()
class MarginL1Loss(L1Loss):
    """Named by the config (``loss_center``) but never built by the head
    (centerpoint_head_gga.py:84 is commented out); registered so the
    unchanged config dict is accepted."""
