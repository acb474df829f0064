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


# ----------------------------------------------------------------------------- mono3d (PGD / FCOS3D) losses
# mmdet's FocalLoss / SmoothL1Loss / CrossEntropyLoss / GIoULoss as configs/gga/gga_pdg.py builds them
# (third-party, restated from mmdet 2.2x's published code - parity unpinned like the two above) and the
# in-tree UncertainSmoothL1Loss (mmdet3d/models/losses/uncertain_smooth_l1_loss.py:10-121).
def _weighted(loss, weight, reduction, avg_factor):
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class FocalLoss(nn.Module):
    """Sigmoid focal loss on [N, C] logits with integer targets: a class index in [0, C), anything else = background."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0, activated=False):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.gamma, self.alpha, self.reduction, self.loss_weight, self.activated = gamma, alpha, reduction, loss_weight, activated

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override or self.reduction
        num_classes = pred.size(1)
        # class c is positive where target == c; any other label (C, or FCAF3D's -1) is background - what mmcv's op computes
        t = (target.view(-1, 1) == torch.arange(num_classes, device=pred.device).view(1, -1)).type_as(pred)
        p = pred if self.activated else pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        focal_weight = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        bce = (torch.nn.functional.binary_cross_entropy(pred, t, reduction='none') if self.activated else
               torch.nn.functional.binary_cross_entropy_with_logits(pred, t, reduction='none'))
        loss = bce * focal_weight
        if weight is not None and weight.shape != loss.shape:
            weight = weight.view(-1, 1) if weight.size(0) == loss.size(0) else weight.view(loss.size(0), -1)
        return self.loss_weight * _weighted(loss, weight, reduction, avg_factor)


def smooth_l1(pred, target, beta):
    diff = torch.abs(pred - target)
    return torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override or self.reduction
        if target.numel() == 0:
            return self.loss_weight * (pred.sum() * 0)
        return self.loss_weight * _weighted(smooth_l1(pred, target, self.beta), weight, reduction, avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """``use_sigmoid=False``: softmax cross entropy on [N, C] logits / integer labels;
    ``use_sigmoid=True``: binary cross entropy with logits against float targets of the same shape
    (the centerness branch)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, ignore_index=None,
                 loss_weight=1.0, avg_non_ignore=False):
        super().__init__()
        assert not use_mask
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight
        self.class_weight, self.ignore_index, self.avg_non_ignore = class_weight, ignore_index, avg_non_ignore

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=None, **kwargs):
        reduction = reduction_override or self.reduction
        ignore_index = -100 if (ignore_index if ignore_index is not None else self.ignore_index) is None else \
            (ignore_index if ignore_index is not None else self.ignore_index)
        cw = cls_score.new_tensor(self.class_weight) if self.class_weight is not None else None
        if self.use_sigmoid:
            assert cls_score.dim() == label.dim(), 'the mono3d heads only use same-shape binary targets'
            valid = ((label >= 0) & (label != ignore_index)).float()
            weight = weight * valid if weight is not None else valid
            if avg_factor is None and self.avg_non_ignore and reduction == 'mean':
                avg_factor = valid.sum().item()
            loss = torch.nn.functional.binary_cross_entropy_with_logits(cls_score, label.float(), pos_weight=cw, reduction='none')
            return self.loss_weight * _weighted(loss, weight.float(), reduction, avg_factor)
        loss = torch.nn.functional.cross_entropy(cls_score, label, weight=cw, reduction='none', ignore_index=ignore_index)
        if avg_factor is None and self.avg_non_ignore and reduction == 'mean':
            avg_factor = label.numel() - (label == ignore_index).sum().item()
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * _weighted(loss, weight, reduction, avg_factor)


def aligned_giou(pred, target, eps=1e-7):
    """mmdet bbox_overlaps(mode='giou', is_aligned=True) for [N, 4] (x1, y1, x2, y2)."""
    area1 = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    area2 = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    lt = torch.max(pred[:, :2], target[:, :2])
    rb = torch.min(pred[:, 2:], target[:, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    union = area1 + area2 - overlap
    enclosed_lt = torch.min(pred[:, :2], target[:, :2])
    enclosed_rb = torch.max(pred[:, 2:], target[:, 2:])
    eps_t = union.new_tensor([eps])
    union = torch.max(union, eps_t)
    ious = overlap / union
    enclose_wh = (enclosed_rb - enclosed_lt).clamp(min=0)
    enclose_area = torch.max(enclose_wh[:, 0] * enclose_wh[:, 1], eps_t)
    return ious - (enclose_area - union) / enclose_area


@LOSSES.register_module()
class GIoULoss(nn.Module):
    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if weight is not None and not torch.any(weight > 0):
            if pred.dim() == weight.dim() + 1:
                weight = weight.unsqueeze(1)
            return (pred * weight).sum()
        reduction = reduction_override or self.reduction
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        loss = 1 - aligned_giou(pred, target, self.eps)
        return self.loss_weight * _weighted(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class UncertainSmoothL1Loss(nn.Module):
    """exp(-sigma) * smooth_l1(pred, target) + alpha * sigma (uncertain_smooth_l1_loss.py:10-38, :62-121)."""

    def __init__(self, alpha=1.0, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert reduction in ['none', 'sum', 'mean']
        self.alpha, self.beta, self.reduction, self.loss_weight = alpha, beta, reduction, loss_weight

    def forward(self, pred, target, sigma, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override or self.reduction
        assert self.beta > 0 and target.numel() > 0 and pred.size() == target.size() == sigma.size()
        loss = torch.exp(-sigma) * smooth_l1(pred, target, self.beta) + self.alpha * sigma
        return self.loss_weight * _weighted(loss, weight, reduction, avg_factor)
