"""FCAF3D (BASELINE config 4): ``MinkResNet`` backbone, ``FCAF3DHead`` (sparse FPN with generative transposed convolutions
and pruning + anchor-free head) and ``MinkSingleStage3DDetector`` - registry names, constructor keys, parameter names and
loss-dict keys of the reference (mmdet3d/models/backbones/mink_resnet.py:18-114, mmdet3d/models/dense_heads/
fcaf3d_head.py:49-678, mmdet3d/models/detectors/mink_single_stage.py:16-109; configs/_base_/models/fcaf3d.py,
configs/fcaf3d/fcaf3d_8x2_sunrgbd-3d-10class.py), on the MinkowskiEngine-semantics layers of ``gga_amd.mink``.

There is no GGA head for this trunk in the reference tree: what is built - and what can be checked - is
stock FCAF3D. Third-party pieces restated from their published code, parity unpinned: MinkowskiEngine (``mink.py``), mmcv's
``diff_iou_rotated_3d`` (``rotated_iou_3d`` below), ``nms3d`` / ``nms3d_normal`` (BEV overlap of the upright boxes, on
``postproc.hip``'s rotated-NMS kernel), ``Scale``, ``bias_init_with_prob``; mmdet's ``FocalLoss`` /
``CrossEntropyLoss`` are those of ``losses.py``. The head's own target assignment, loss and box decoding are plain torch in the
reference and are pinned by golden vectors from a run of the reference's class (tests/golden/fcaf3d_head.npz)."""
import math

import torch
from torch import nn

from . import mink as ME
from . import ops
from .losses import weight_reduce_loss
from .registry import BACKBONES, DETECTORS, HEADS, LOSSES, build_backbone, build_head, build_loss


# ----------------------------------------------------------------------------------------------------------------- boxes
def rotation_3d_in_axis_z(points, angles):
    """``rotation_3d_in_axis(points [N, M, 3], angles [N], axis=2)`` (core/bbox/structures/utils.py:79-117): points @ R with
    R = [[cos, sin, 0], [-sin, cos, 0], [0, 0, 1]] per entry - the counter-clockwise rotation of the row vectors."""
    c, s = torch.cos(angles), torch.sin(angles)
    o, z = torch.ones_like(c), torch.zeros_like(c)
    rot = torch.stack([torch.stack([c, s, z], -1), torch.stack([-s, c, z], -1), torch.stack([z, z, o], -1)], -2)     # [N, 3, 3]
    return torch.einsum('aij,ajk->aik', points, rot)


class DepthInstance3DBoxes:
    """The part of ``DepthInstance3DBoxes`` (core/bbox/structures/depth_box3d.py) the FCAF3D head touches: (x, y, z, dx, dy,
    dz[, yaw]) with the origin converted to the bottom centre (0.5, 0.5, 0)."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim))
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim, tensor.size()
        if tensor.shape[-1] == 6:
            assert box_dim == 6
            tensor = torch.cat((tensor, tensor.new_zeros(tensor.shape[0], 1)), dim=-1)
            self.box_dim, self.with_yaw = box_dim + 1, False
        else:
            self.box_dim, self.with_yaw = box_dim, with_yaw
        self.tensor = tensor.clone()
        if origin != (0.5, 0.5, 0):
            dst, src = self.tensor.new_tensor((0.5, 0.5, 0)), self.tensor.new_tensor(origin)
            self.tensor[:, :3] += self.tensor[:, 3:6] * (dst - src)

    @property
    def gravity_center(self):
        bc = self.tensor[:, :3]
        gc = torch.zeros_like(bc)
        gc[:, :2] = bc[:, :2]
        gc[:, 2] = bc[:, 2] + self.tensor[:, 5] * 0.5
        return gc

    volume = property(lambda self: self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5])

    def __len__(self):
        return self.tensor.shape[0]

    def to(self, device):
        out = object.__new__(type(self))
        out.tensor, out.box_dim, out.with_yaw = self.tensor.to(device), self.box_dim, self.with_yaw
        return out


# ---------------------------------------------------------------------------------------------------------------- losses
def axis_aligned_iou(b1, b2, eps=1e-6):
    """``axis_aligned_bbox_overlaps_3d(..., mode='iou', is_aligned=True)`` (iou3d_calculator.py:210-329) on (x1,y1,z1,x2,y2,z2)."""
    a1 = (b1[..., 3] - b1[..., 0]) * (b1[..., 4] - b1[..., 1]) * (b1[..., 5] - b1[..., 2])
    a2 = (b2[..., 3] - b2[..., 0]) * (b2[..., 4] - b2[..., 1]) * (b2[..., 5] - b2[..., 2])
    wh = (torch.min(b1[..., 3:], b2[..., 3:]) - torch.max(b1[..., :3], b2[..., :3])).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1] * wh[..., 2]
    union = torch.max(a1 + a2 - overlap, a1.new_tensor([eps]))
    return overlap / union


def _box2corners(box):
    """(..., 5) x, y, w, h, alpha -> (..., 4, 2), mmcv's corner order."""
    x, y, w, h, alpha = box.split([1, 1, 1, 1, 1], dim=-1)
    x4 = box.new_tensor([0.5, -0.5, -0.5, 0.5]) * w
    y4 = box.new_tensor([0.5, 0.5, -0.5, -0.5]) * h
    sin, cos = torch.sin(alpha), torch.cos(alpha)
    return torch.stack([x4 * cos - y4 * sin + x, x4 * sin + y4 * cos + y], dim=-1)


def _in_box(c1, c2):
    a, b, d = c2[..., 0:1, :], c2[..., 1:2, :], c2[..., 3:4, :]
    ab, am, ad = b - a, c1 - a, d - a
    p1, p2 = (ab * am).sum(-1) / (ab * ab).sum(-1), (ad * am).sum(-1) / (ad * ad).sum(-1)
    return (p1 > -1e-6) & (p1 < 1 + 1e-6) & (p2 > -1e-6) & (p2 < 1 + 1e-6)


def rotated_iou_3d(box1, box2):
    """mmcv ``diff_iou_rotated_3d`` for aligned pairs, [N, 7] (x, y, z, w, h, l, alpha) -> IoU [N], differentiable: the 24
    candidate vertices of the BEV intersection polygon (4 + 4 corners, 16 edge intersections), the valid ones ordered by
    angle around their mean (mmcv does this step in a CUDA op; indices only, no gradient), shoelace area, times the
    overlap in z."""
    c1, c2 = _box2corners(box1[:, [0, 1, 3, 4, 6]]), _box2corners(box2[:, [0, 1, 3, 4, 6]])          # [N, 4, 2]
    l1 = torch.cat([c1, c1[:, [1, 2, 3, 0]]], -1)[:, :, None]                                       # [N, 4, 1, 4]
    l2 = torch.cat([c2, c2[:, [1, 2, 3, 0]]], -1)[:, None]                                          # [N, 1, 4, 4]
    x1, y1, x2, y2 = l1.unbind(-1)
    x3, y3, x4, y4 = l2.unbind(-1)
    num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)
    den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)
    zero = num == 0
    t = torch.where(zero, torch.full_like(num, -1.0), den_t / torch.where(zero, torch.ones_like(num), num))
    u = torch.where(zero, torch.full_like(num, -1.0), -den_u / torch.where(zero, torch.ones_like(num), num))
    mask = ((t > 0) & (t < 1) & (u > 0) & (u < 1)).detach()
    t = den_t / (num + 1e-8)
    inter = torch.stack([x1 + t * (x2 - x1), y1 + t * (y2 - y1)], -1) * mask[..., None].float()    # [N, 4, 4, 2]
    N = box1.shape[0]
    verts = torch.cat([c1, c2, inter.reshape(N, 16, 2), c1.new_zeros(N, 1, 2)], 1)                  # [N, 25, 2]; 24 = the zero pad vertex
    vmask = torch.cat([_in_box(c1, c2), _in_box(c2, c1), mask.reshape(N, 16), mask.new_zeros(N, 1)], 1)
    with torch.no_grad():
        nv = vmask.sum(1)
        mean = (verts * vmask[..., None].float()).sum(1, keepdim=True) / nv.clamp(min=1)[:, None, None]
        vn = verts - mean
        ang = torch.atan2(vn[..., 1], vn[..., 0])
        ang = torch.where(vmask, ang, torch.full_like(ang, float('inf')))
        order = torch.argsort(ang, dim=1)[:, :8]                                                    # valid vertices first, by angle
        j = torch.arange(9, device=box1.device)[None]
        idx = torch.where(j < nv[:, None], torch.cat([order, order[:, :1]], 1), torch.full((N, 9), 24, device=box1.device))
        idx = torch.where(j == nv[:, None], order[:, :1].expand(-1, 9), idx)                        # close the polygon
        idx = torch.where((nv < 3)[:, None], torch.full_like(idx, 24), idx)
    sel = torch.gather(verts, 1, idx[..., None].expand(-1, -1, 2))
    area = (sel[:, :-1, 0] * sel[:, 1:, 1] - sel[:, :-1, 1] * sel[:, 1:, 0]).sum(1).abs() / 2
    zmax = torch.min(box1[:, 2] + box1[:, 5] * 0.5, box2[:, 2] + box2[:, 5] * 0.5)
    zmin = torch.max(box1[:, 2] - box1[:, 5] * 0.5, box2[:, 2] - box2[:, 5] * 0.5)
    inter3d = area * (zmax - zmin).clamp(min=0)
    vol1, vol2 = box1[:, 3] * box1[:, 4] * box1[:, 5], box2[:, 3] * box2[:, 4] * box2[:, 5]
    return inter3d / (vol1 + vol2 - inter3d)


class _IoULoss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert reduction in ('none', 'sum', 'mean')
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and not torch.any(weight > 0):          # the two reference classes word this case differently
            if type(self) is RotatedIoU3DLoss:
                return pred.sum() * weight.sum()
            if reduction != 'none':
                return (pred * weight).sum()
        if type(self) is RotatedIoU3DLoss and weight is not None and weight.dim() > 1:
            weight = weight.mean(-1)
        return self.loss_weight * weight_reduce_loss(1 - self.iou(pred, target), weight, reduction, avg_factor)


@LOSSES.register_module()
class AxisAlignedIoULoss(_IoULoss):
    """models/losses/axis_aligned_iou_loss.py:10-80: 1 - IoU of (x1, y1, z1, x2, y2, z2) boxes."""
    iou = staticmethod(axis_aligned_iou)


@LOSSES.register_module()
class RotatedIoU3DLoss(_IoULoss):
    """models/losses/rotated_iou_loss.py:10-86: 1 - diff_iou_rotated_3d of (x, y, z, w, l, h, alpha) boxes."""
    iou = staticmethod(rotated_iou_3d)


def nms3d(boxes, scores, iou_threshold):
    """mmcv ``nms3d``: greedy NMS on the BEV overlap of the upright boxes [N, 7] (x, y, z, dx, dy, dz, heading) -> kept
    indices in descending score order."""
    if boxes.shape[0] == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    return ops.nms_rotated(boxes[:, [0, 1, 3, 4, 6]].contiguous(), scores.contiguous(), iou_threshold)[1]


def nms3d_normal(boxes, scores, iou_threshold):
    """mmcv ``nms3d_normal``: the same with the heading ignored (axis-aligned BEV boxes)."""
    b = boxes[:, [0, 1, 3, 4, 6]].clone()
    b[:, 4] = 0
    if boxes.shape[0] == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    return ops.nms_rotated(b.contiguous(), scores.contiguous(), iou_threshold)[1]


class _SplitRows(torch.autograd.Function):
    """t [n, C], index vectors with DISTINCT rows overall -> tuple(t[p] for p in perms). Backward: every gradient is copied
    to its rows of one zero tensor (no duplicates, so no accumulation and no sort)."""

    @staticmethod
    def forward(ctx, t, *perms):
        ctx.perms, ctx.shape = perms, t.shape
        return tuple(t.index_select(0, p) for p in perms)

    @staticmethod
    def backward(ctx, *gs):
        g = gs[0].new_zeros(ctx.shape) if gs[0] is not None else None
        for p, gi in zip(ctx.perms, gs):
            if gi is not None:
                if g is None:
                    g = gi.new_zeros(ctx.shape)
                g.index_copy_(0, p, gi)
        return (g,) + (None,) * len(ctx.perms)


def split_rows(t, perms):
    """[t[p] for p in perms] for index vectors that name every row at most once (the samples of a batch)."""
    perms = list(perms)
    if not perms or not t.requires_grad:
        return [t[p] for p in perms]
    return list(_SplitRows.apply(t, *perms))


def take_rows(t, idx):
    """t[idx] for an index vector without duplicates (e.g. from nonzero)."""
    if not t.requires_grad:
        return t[idx]
    return _SplitRows.apply(t, idx)[0]


class Scale(nn.Module):
    """mmcv.cnn.Scale: a learnable scalar factor."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def bias_init_with_prob(prior_prob):
    return float(-math.log((1 - prior_prob) / prior_prob))


def reduce_mean(tensor):
    """mmdet.core.reduce_mean: the mean over the ranks (the one collective of this head besides the gradient all-reduce)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor


# -------------------------------------------------------------------------------------------------------------- backbone
@BACKBONES.register_module()
class MinkResNet(nn.Module):
    arch_settings = {18: (ME.BasicBlock, (2, 2, 2, 2)), 34: (ME.BasicBlock, (3, 4, 6, 3))}

    def __init__(self, depth, in_channels, num_stages=4, pool=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet (the BasicBlock depths 18 / 34 are built)')
        assert 4 >= num_stages >= 1
        block, stage_blocks = self.arch_settings[depth]
        stage_blocks = stage_blocks[:num_stages]
        self.num_stages, self.pool = num_stages, pool
        self.inplanes = 64
        self.conv1 = ME.MinkowskiConvolution(in_channels, self.inplanes, kernel_size=3, stride=2, dimension=3)
        self.norm1 = ME.MinkowskiInstanceNorm(self.inplanes)
        self.relu = ME.MinkowskiReLU(inplace=True)
        if self.pool:
            self.maxpool = ME.MinkowskiMaxPooling(kernel_size=2, stride=2, dimension=3)
        for i, _ in enumerate(stage_blocks):
            setattr(self, f'layer{i + 1}', self._make_layer(block, 64 * 2 ** i, stage_blocks[i], stride=2))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, ME.MinkowskiConvolution):        # ME.utils.kaiming_normal_(kernel, mode='fan_out', nonlinearity='relu')
                fan_out = m.kernel.shape[0] * m.kernel.shape[2]
                nn.init.normal_(m.kernel, 0, math.sqrt(2.0 / fan_out))
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                ME.MinkowskiConvolution(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, dimension=3),
                ME.MinkowskiBatchNorm(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride=stride, downsample=downsample, dimension=3)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, stride=1, dimension=3))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.relu(self.norm1(self.conv1(x)))
        if self.pool:
            x = self.maxpool(x)
        outs = []
        for i in range(self.num_stages):
            x = getattr(self, f'layer{i + 1}')(x)
            outs.append(x)
        return outs


# ------------------------------------------------------------------------------------------------------------------ head
@HEADS.register_module()
class FCAF3DHead(nn.Module):
    def __init__(self, n_classes, in_channels, out_channels, n_reg_outs, voxel_size, pts_prune_threshold, pts_assign_threshold,
                 pts_center_threshold, center_loss=dict(type='CrossEntropyLoss', use_sigmoid=True),
                 bbox_loss=dict(type='AxisAlignedIoULoss'), cls_loss=dict(type='FocalLoss'), train_cfg=None, test_cfg=None,
                 init_cfg=None):
        super().__init__()
        self.voxel_size = voxel_size
        self.pts_prune_threshold, self.pts_assign_threshold, self.pts_center_threshold = pts_prune_threshold, pts_assign_threshold, pts_center_threshold
        self.center_loss, self.bbox_loss, self.cls_loss = build_loss(center_loss), build_loss(bbox_loss), build_loss(cls_loss)
        from .config import ConfigDict
        self.train_cfg = train_cfg
        self.test_cfg = ConfigDict(test_cfg) if isinstance(test_cfg, dict) and not isinstance(test_cfg, ConfigDict) else test_cfg
        self._init_layers(in_channels, out_channels, n_reg_outs, n_classes)

    @staticmethod
    def _make_block(in_channels, out_channels):
        return nn.Sequential(ME.MinkowskiConvolution(in_channels, out_channels, kernel_size=3, dimension=3),
                             ME.MinkowskiBatchNorm(out_channels), ME.MinkowskiELU())

    @staticmethod
    def _make_up_block(in_channels, out_channels):
        return nn.Sequential(
            ME.MinkowskiGenerativeConvolutionTranspose(in_channels, out_channels, kernel_size=2, stride=2, dimension=3),
            ME.MinkowskiBatchNorm(out_channels), ME.MinkowskiELU(),
            ME.MinkowskiConvolution(out_channels, out_channels, kernel_size=3, dimension=3),
            ME.MinkowskiBatchNorm(out_channels), ME.MinkowskiELU())

    def _init_layers(self, in_channels, out_channels, n_reg_outs, n_classes):
        self.pruning = ME.MinkowskiPruning()
        for i in range(len(in_channels)):
            if i > 0:
                self.__setattr__(f'up_block_{i}', self._make_up_block(in_channels[i], in_channels[i - 1]))
            self.__setattr__(f'out_block_{i}', self._make_block(in_channels[i], out_channels))
        self.conv_center = ME.MinkowskiConvolution(out_channels, 1, kernel_size=1, dimension=3)
        self.conv_reg = ME.MinkowskiConvolution(out_channels, n_reg_outs, kernel_size=1, dimension=3)
        self.conv_cls = ME.MinkowskiConvolution(out_channels, n_classes, kernel_size=1, bias=True, dimension=3)
        self.scales = nn.ModuleList([Scale(1.) for _ in range(len(in_channels))])

    def init_weights(self):
        nn.init.normal_(self.conv_center.kernel, std=.01)
        nn.init.normal_(self.conv_reg.kernel, std=.01)
        nn.init.normal_(self.conv_cls.kernel, std=.01)
        nn.init.constant_(self.conv_cls.bias, bias_init_with_prob(.01))

    def forward(self, x):
        center_preds, bbox_preds, cls_preds, points = [], [], [], []
        inputs = x
        x = inputs[-1]
        prune_score = None
        for i in range(len(inputs) - 1, -1, -1):
            if i < len(inputs) - 1:
                x = self.__getattr__(f'up_block_{i + 1}')(x)
                x = inputs[i] + x
                x = self._prune(x, prune_score)
            out = self.__getattr__(f'out_block_{i}')(x)
            center_pred, bbox_pred, cls_pred, point, prune_score = self._forward_single(out, self.scales[i])
            center_preds.append(center_pred)
            bbox_preds.append(bbox_pred)
            cls_preds.append(cls_pred)
            points.append(point)
        return center_preds[::-1], bbox_preds[::-1], cls_preds[::-1], points[::-1]

    def forward_train(self, x, gt_bboxes, gt_labels, input_metas):
        center_preds, bbox_preds, cls_preds, points = self(x)
        return self._loss(center_preds, bbox_preds, cls_preds, points, gt_bboxes, gt_labels, input_metas)

    def forward_test(self, x, input_metas):
        center_preds, bbox_preds, cls_preds, points = self(x)
        return self._get_bboxes(center_preds, bbox_preds, cls_preds, points, input_metas)

    def _prune(self, x, scores):
        with torch.no_grad():
            coordinates = x.C.float()
            interpolated_scores = scores.features_at_coordinates(coordinates)
            prune_mask = interpolated_scores.new_zeros((len(interpolated_scores)), dtype=torch.bool)
            for permutation in x.decomposition_permutations:
                score = interpolated_scores[permutation]
                mask = score.new_zeros((len(score)), dtype=torch.bool)
                topk = min(len(score), self.pts_prune_threshold)
                ids = torch.topk(score.squeeze(1), topk, sorted=False).indices
                mask[ids] = True
                prune_mask[permutation[mask]] = True
        return self.pruning(x, prune_mask)

    def _forward_single(self, x, scale):
        center_pred = self.conv_center(x).features
        scores = self.conv_cls(x)
        cls_pred = scores.features
        prune_scores = ME.SparseTensor(scores.features.max(dim=1, keepdim=True).values, cmap=scores.cmap)
        reg_final = self.conv_reg(x).features
        reg_distance = torch.exp(scale(reg_final[:, :6]))
        reg_angle = reg_final[:, 6:]
        bbox_pred = torch.cat((reg_distance, reg_angle), dim=1)
        # per-sample rows (the reference indexes with every permutation in turn: x[permutation] - whose backward in torch is a
        # sort-based scatter of the whole tensor per use, 83 launches of 0.6 ms per step here; the permutations partition the rows)
        perms = x.decomposition_permutations
        center_preds, bbox_preds, cls_preds = split_rows(center_pred, perms), split_rows(bbox_pred, perms), split_rows(cls_pred, perms)
        points = x.decomposed_coordinates
        for i in range(len(points)):
            points[i] = points[i] * self.voxel_size
        return center_preds, bbox_preds, cls_preds, points, prune_scores

    def _loss_single(self, center_preds, bbox_preds, cls_preds, points, gt_bboxes, gt_labels, input_meta):
        center_targets, bbox_targets, cls_targets = self._get_targets(points, gt_bboxes, gt_labels)
        center_preds, bbox_preds, cls_preds, points = torch.cat(center_preds), torch.cat(bbox_preds), torch.cat(cls_preds), torch.cat(points)
        pos_inds = torch.nonzero(cls_targets >= 0).squeeze(1)
        n_pos = points.new_tensor(len(pos_inds))
        n_pos = max(reduce_mean(n_pos), 1.)
        cls_loss = self.cls_loss(cls_preds, cls_targets, avg_factor=n_pos)
        pos_center_preds, pos_bbox_preds = take_rows(center_preds, pos_inds), take_rows(bbox_preds, pos_inds)
        pos_center_targets = center_targets[pos_inds].unsqueeze(1)
        pos_bbox_targets = bbox_targets[pos_inds]
        center_denorm = max(reduce_mean(pos_center_targets.sum().detach()), 1e-6)        # outside the branch: no deadlock
        if len(pos_inds) > 0:
            pos_points = points[pos_inds]
            center_loss = self.center_loss(pos_center_preds, pos_center_targets, avg_factor=n_pos)
            bbox_loss = self.bbox_loss(self._bbox_to_loss(self._bbox_pred_to_bbox(pos_points, pos_bbox_preds)),
                                       self._bbox_to_loss(pos_bbox_targets), weight=pos_center_targets.squeeze(1),
                                       avg_factor=center_denorm)
        else:
            center_loss, bbox_loss = pos_center_preds.sum(), pos_bbox_preds.sum()
        return center_loss, bbox_loss, cls_loss

    def _loss(self, center_preds, bbox_preds, cls_preds, points, gt_bboxes, gt_labels, input_metas):
        center_losses, bbox_losses, cls_losses = [], [], []
        for i in range(len(input_metas)):
            center_loss, bbox_loss, cls_loss = self._loss_single(
                center_preds=[x[i] for x in center_preds], bbox_preds=[x[i] for x in bbox_preds], cls_preds=[x[i] for x in cls_preds],
                points=[x[i] for x in points], input_meta=input_metas[i], gt_bboxes=gt_bboxes[i], gt_labels=gt_labels[i])
            center_losses.append(center_loss)
            bbox_losses.append(bbox_loss)
            cls_losses.append(cls_loss)
        return dict(center_loss=torch.mean(torch.stack(center_losses)), bbox_loss=torch.mean(torch.stack(bbox_losses)),
                    cls_loss=torch.mean(torch.stack(cls_losses)))

    def _get_bboxes_single(self, center_preds, bbox_preds, cls_preds, points, input_meta):
        mlvl_bboxes, mlvl_scores = [], []
        for center_pred, bbox_pred, cls_pred, point in zip(center_preds, bbox_preds, cls_preds, points):
            scores = cls_pred.sigmoid() * center_pred.sigmoid()
            max_scores, _ = scores.max(dim=1)
            if len(scores) > self.test_cfg.nms_pre > 0:
                _, ids = max_scores.topk(self.test_cfg.nms_pre)
                bbox_pred, scores, point = bbox_pred[ids], scores[ids], point[ids]
            mlvl_bboxes.append(self._bbox_pred_to_bbox(point, bbox_pred))
            mlvl_scores.append(scores)
        return self._single_scene_multiclass_nms(torch.cat(mlvl_bboxes), torch.cat(mlvl_scores), input_meta)

    def _get_bboxes(self, center_preds, bbox_preds, cls_preds, points, input_metas):
        return [self._get_bboxes_single(center_preds=[x[i] for x in center_preds], bbox_preds=[x[i] for x in bbox_preds],
                                        cls_preds=[x[i] for x in cls_preds], points=[x[i] for x in points],
                                        input_meta=input_metas[i]) for i in range(len(input_metas))]

    @staticmethod
    def _bbox_to_loss(bbox):
        if bbox.shape[-1] != 6:          # the rotated IoU loss takes (x, y, z, w, h, l, heading)
            return bbox
        return torch.stack((bbox[..., 0] - bbox[..., 3] / 2, bbox[..., 1] - bbox[..., 4] / 2, bbox[..., 2] - bbox[..., 5] / 2,
                            bbox[..., 0] + bbox[..., 3] / 2, bbox[..., 1] + bbox[..., 4] / 2, bbox[..., 2] + bbox[..., 5] / 2), dim=-1)

    @staticmethod
    def _bbox_pred_to_bbox(points, bbox_pred):
        if bbox_pred.shape[0] == 0:
            return bbox_pred
        x_center = points[:, 0] + (bbox_pred[:, 1] - bbox_pred[:, 0]) / 2
        y_center = points[:, 1] + (bbox_pred[:, 3] - bbox_pred[:, 2]) / 2
        z_center = points[:, 2] + (bbox_pred[:, 5] - bbox_pred[:, 4]) / 2
        base_bbox = torch.stack([x_center, y_center, z_center, bbox_pred[:, 0] + bbox_pred[:, 1], bbox_pred[:, 2] + bbox_pred[:, 3],
                                 bbox_pred[:, 4] + bbox_pred[:, 5]], -1)
        if bbox_pred.shape[1] == 6:
            return base_bbox
        # rotated case: ..., sin(2a) ln(q), cos(2a) ln(q)
        scale = bbox_pred[:, 0] + bbox_pred[:, 1] + bbox_pred[:, 2] + bbox_pred[:, 3]
        q = torch.exp(torch.sqrt(torch.pow(bbox_pred[:, 6], 2) + torch.pow(bbox_pred[:, 7], 2)))
        alpha = 0.5 * torch.atan2(bbox_pred[:, 6], bbox_pred[:, 7])
        return torch.stack((x_center, y_center, z_center, scale / (1 + q), scale / (1 + q) * q, bbox_pred[:, 5] + bbox_pred[:, 4], alpha), dim=-1)

    @staticmethod
    def _get_face_distances(points, boxes):
        shift = torch.stack((points[..., 0] - boxes[..., 0], points[..., 1] - boxes[..., 1], points[..., 2] - boxes[..., 2]), dim=-1).permute(1, 0, 2)
        shift = rotation_3d_in_axis_z(shift, -boxes[0, :, 6]).permute(1, 0, 2)
        centers = boxes[..., :3] + shift
        dx_min = centers[..., 0] - boxes[..., 0] + boxes[..., 3] / 2
        dx_max = boxes[..., 0] + boxes[..., 3] / 2 - centers[..., 0]
        dy_min = centers[..., 1] - boxes[..., 1] + boxes[..., 4] / 2
        dy_max = boxes[..., 1] + boxes[..., 4] / 2 - centers[..., 1]
        dz_min = centers[..., 2] - boxes[..., 2] + boxes[..., 5] / 2
        dz_max = boxes[..., 2] + boxes[..., 5] / 2 - centers[..., 2]
        return torch.stack((dx_min, dx_max, dy_min, dy_max, dz_min, dz_max), dim=-1)

    @staticmethod
    def _get_centerness(face_distances):
        x_dims, y_dims, z_dims = face_distances[..., [0, 1]], face_distances[..., [2, 3]], face_distances[..., [4, 5]]
        centerness_targets = x_dims.min(dim=-1)[0] / x_dims.max(dim=-1)[0] * y_dims.min(dim=-1)[0] / y_dims.max(dim=-1)[0] * \
            z_dims.min(dim=-1)[0] / z_dims.max(dim=-1)[0]
        return torch.sqrt(centerness_targets)

    @torch.no_grad()
    def _get_targets(self, points, gt_bboxes, gt_labels):
        float_max = points[0].new_tensor(1e8)
        n_levels = len(points)
        levels = torch.cat([points[i].new_tensor(i).expand(len(points[i])) for i in range(len(points))])
        points = torch.cat(points)
        gt_bboxes = gt_bboxes.to(points.device)
        n_points, n_boxes = len(points), len(gt_bboxes)
        volumes = gt_bboxes.volume.unsqueeze(0).expand(n_points, n_boxes)
        # condition 1: point inside box
        boxes = torch.cat((gt_bboxes.gravity_center, gt_bboxes.tensor[:, 3:]), dim=1)
        boxes = boxes.expand(n_points, n_boxes, 7)
        points = points.unsqueeze(1).expand(n_points, n_boxes, 3)
        face_distances = self._get_face_distances(points, boxes)
        inside_box_condition = face_distances.min(dim=-1).values > 0
        # condition 2: positive points per level >= limit
        n_pos_points_per_level = torch.stack([torch.sum(inside_box_condition[levels == i], dim=0) for i in range(n_levels)], dim=0)
        lower_limit_mask = n_pos_points_per_level < self.pts_assign_threshold
        lower_index = torch.argmax(lower_limit_mask.int(), dim=0) - 1
        lower_index = torch.where(lower_index < 0, 0, lower_index)
        all_upper_limit_mask = torch.all(torch.logical_not(lower_limit_mask), dim=0)
        best_level = torch.where(all_upper_limit_mask, n_levels - 1, lower_index)
        best_level = best_level.expand(n_points, n_boxes)
        levels = torch.unsqueeze(levels, 1).expand(n_points, n_boxes)
        level_condition = best_level == levels
        # condition 3: limit topk points per box by centerness
        centerness = self._get_centerness(face_distances)
        centerness = torch.where(inside_box_condition, centerness, torch.ones_like(centerness) * -1)
        centerness = torch.where(level_condition, centerness, torch.ones_like(centerness) * -1)
        top_centerness = torch.topk(centerness, min(self.pts_center_threshold + 1, len(centerness)), dim=0).values[-1]
        topk_condition = centerness > top_centerness.unsqueeze(0)
        # condition 4: min volume box per point
        volumes = torch.where(inside_box_condition, volumes, float_max)
        volumes = torch.where(level_condition, volumes, float_max)
        volumes = torch.where(topk_condition, volumes, float_max)
        min_volumes, min_inds = volumes.min(dim=1)
        center_targets = centerness[torch.arange(n_points), min_inds]
        bbox_targets = boxes[torch.arange(n_points), min_inds]
        if not gt_bboxes.with_yaw:
            bbox_targets = bbox_targets[:, :-1]
        cls_targets = gt_labels[min_inds]
        cls_targets = torch.where(min_volumes == float_max, -1, cls_targets)
        return center_targets, bbox_targets, cls_targets

    def _single_scene_multiclass_nms(self, bboxes, scores, input_meta):
        n_classes = scores.shape[1]
        with_yaw = bboxes.shape[1] == 7
        nms_bboxes, nms_scores, nms_labels = [], [], []
        for i in range(n_classes):
            ids = scores[:, i] > self.test_cfg.score_thr
            if not ids.any():
                continue
            class_scores, class_bboxes = scores[ids, i], bboxes[ids]
            if with_yaw:
                nms_function = nms3d
            else:
                class_bboxes = torch.cat((class_bboxes, torch.zeros_like(class_bboxes[:, :1])), dim=1)
                nms_function = nms3d_normal
            nms_ids = nms_function(class_bboxes, class_scores, self.test_cfg.iou_thr)
            nms_bboxes.append(class_bboxes[nms_ids])
            nms_scores.append(class_scores[nms_ids])
            nms_labels.append(bboxes.new_full(class_scores[nms_ids].shape, i, dtype=torch.long))
        if len(nms_bboxes):
            nms_bboxes, nms_scores, nms_labels = torch.cat(nms_bboxes, dim=0), torch.cat(nms_scores, dim=0), torch.cat(nms_labels, dim=0)
        else:
            nms_bboxes, nms_scores, nms_labels = bboxes.new_zeros((0, bboxes.shape[1])), bboxes.new_zeros((0, )), bboxes.new_zeros((0, ))
        if with_yaw:
            box_dim = 7
        else:
            box_dim = 6
            nms_bboxes = nms_bboxes[:, :6]
        nms_bboxes = input_meta['box_type_3d'](nms_bboxes, box_dim=box_dim, with_yaw=with_yaw, origin=(.5, .5, .5))
        return nms_bboxes, nms_scores, nms_labels


# -------------------------------------------------------------------------------------------------------------- detector
@DETECTORS.register_module()
class MinkSingleStage3DDetector(nn.Module):
    def __init__(self, backbone, head, voxel_size, train_cfg=None, test_cfg=None, init_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        head = dict(head)
        head.update(train_cfg=train_cfg)
        head.update(test_cfg=test_cfg)
        self.head = build_head(head)
        self.voxel_size = voxel_size
        self.init_weights()

    def init_weights(self):
        self.backbone.init_weights()
        self.head.init_weights()

    def extract_feat(self, points):
        coordinates, features = ME.batch_sparse_collate([(p[:, :3] / self.voxel_size, p[:, 3:]) for p in points], device=points[0].device)
        x = ME.SparseTensor(coordinates=coordinates, features=features, batch_size=len(points))
        return self.backbone(x)

    def forward_train(self, points, gt_bboxes_3d, gt_labels_3d, img_metas):
        x = self.extract_feat(points)
        return self.head.forward_train(x, gt_bboxes_3d, gt_labels_3d, img_metas)

    @torch.no_grad()
    def simple_test(self, points, img_metas, *args, **kwargs):
        from .box3d import bbox3d2result
        x = self.extract_feat(points)
        bbox_list = self.head.forward_test(x, img_metas)
        return [bbox3d2result(bboxes, scores, labels) for bboxes, scores, labels in bbox_list]

    def aug_test(self, points, img_metas, **kwargs):
        raise NotImplementedError

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.simple_test(kwargs['points'][0], kwargs['img_metas'][0])

    def _parse_losses(self, losses):
        from collections import OrderedDict
        log_vars = OrderedDict((k, v.mean() if isinstance(v, torch.Tensor) else sum(x.mean() for x in v)) for k, v in losses.items())
        loss = torch.stack([v for k, v in log_vars.items() if 'loss' in k]).sum()
        log_vars['loss'] = loss
        return loss, log_vars

    def train_step(self, data, optimizer=None):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))
