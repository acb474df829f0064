"""``PGDHead`` with its bases ``FCOSMono3DHead`` / ``AnchorFreeMono3DHead`` - SURVEY.md §8(f) rank 4, the
camera-only detector the GGA recipe retrains on its pseudo labels (configs/gga/gga_pdg.py ->
configs/_base_/models/pgd.py). Constructor arguments, layer / parameter names (``cls_convs``, ``reg_convs``,
``conv_cls_prev``, ``conv_regs``, ``scales``, ``fuse_lambda`` ...), the forward outputs and the loss-dict
keys are the reference's (mmdet3d/models/dense_heads/anchor_free_mono3d_head.py:15-534,
fcos_mono3d_head.py:20-956, pgd_head.py:17-1229).

MI355X side: the last tower convolutions are ``DCNv2`` (gga_amd/dcn.py, HIP sampling kernels); the
O(points x boxes) target assignment of the whole batch is ONE launch (``gga_fcos3d_targets``) instead
of ~60 broadcast tensor ops per image in a Python loop; the loss arithmetic on the few hundred
positive points is plain device tensor code in the reference's order. Inference (``get_bboxes``, pgd_head.py:878-1130):
per-level top-k, decoding, then per-class BEV NMS on the rotated-NMS kernel (``ops.box3d_multiclass_nms``).
"""
import math

import numpy as np
import os
import torch
from torch import nn

from . import _lib
from . import functional as F
from ._lib import check
from .bbox_coders import limit_period
from .box3d import points_cam2img, points_img2cam
from .cnn import ConvModule
from .registry import HEADS, build_bbox_coder, build_loss

INF = 1e8


class _ShareParams(torch.autograd.Function):
    """``n`` aliases of every parameter (level-major: alias c of parameter i is output c * P + i). Backward: the aliases'
    gradients added with multi-tensor launches (a gradient that never arrived - a parameter one level does not use - is
    skipped), on the stream this node was made on."""

    @staticmethod
    def forward(ctx, n, *params):
        ctx.n, ctx.P = n, len(params)
        ctx.set_materialize_grads(False)       # an alias nobody differentiated through stays None (no zero tensors, no zero gradients)
        return tuple(p.detach() for _ in range(n) for p in params)

    @staticmethod
    def backward(ctx, *gs):
        n, P = ctx.n, ctx.P
        total = [None] * P
        for c in range(n):
            have = [i for i in range(P) if gs[c * P + i] is not None]
            first = [i for i in have if total[i] is None]
            more = [i for i in have if total[i] is not None]
            if more:
                summed = torch._foreach_add([total[i] for i in more], [gs[c * P + i] for i in more])
                for i, t in zip(more, summed):
                    total[i] = t
            for i in first:
                total[i] = gs[c * P + i]
        return (None,) + tuple(total)


class Scale(nn.Module):
    """mmcv.cnn.Scale: a learnable scalar factor."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def normal_init(module, mean=0, std=1, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))


def distance2bbox(points, distance, max_shape=None):
    """mmdet.core.distance2bbox: (left, top, right, bottom) distances -> xyxy, optionally clamped to the image."""
    x1, y1 = points[..., 0] - distance[..., 0], points[..., 1] - distance[..., 1]
    x2, y2 = points[..., 0] + distance[..., 2], points[..., 1] + distance[..., 3]
    if max_shape is not None:
        x1, x2 = x1.clamp(min=0, max=max_shape[1]), x2.clamp(min=0, max=max_shape[1])
        y1, y2 = y1.clamp(min=0, max=max_shape[0]), y2.clamp(min=0, max=max_shape[0])
    return torch.stack([x1, y1, x2, y2], -1)


_LEVEL_STREAM_CACHE = {}
LEVEL_STREAMS = True      # PGDHead.forward: one stream per FPN level (when LEVEL_BATCH is off)
LEVEL_BATCH = False       # PGDHead.forward: layer by layer over all levels, weight-sharing convolutions as one launch over the
                          # maps (measured: 145 ms per head forward + backward against 135 with LEVEL_STREAMS - the largest
                          # level is three quarters of the work and fills the chip alone)


def multi_apply(func, *args, **kwargs):
    results = [func(*a, **kwargs) for a in zip(*args)]
    return tuple(map(list, zip(*results)))


class AnchorFreeMono3DHead(nn.Module):
    _version = 1

    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=4, strides=(4, 8, 16, 32, 64),
                 dcn_on_last_conv=False, conv_bias='auto', background_label=None, use_direction_classifier=True,
                 diff_rad_by_sin=True, dir_offset=0, dir_limit_offset=0,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_dir=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_attr=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0), bbox_code_size=9,
                 pred_attrs=False, num_attrs=9, pred_velo=False, pred_bbox2d=False, group_reg_dims=(2, 1, 3, 1, 2),
                 cls_branch=(128, 64), reg_branch=((128, 64), (128, 64), (64, ), (64, ), ()), dir_branch=(64, ),
                 attr_branch=(64, ), conv_cfg=None, norm_cfg=None, train_cfg=None, test_cfg=None, init_cfg=None):
        super().__init__()
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels, self.stacked_convs, self.strides = in_channels, feat_channels, stacked_convs, strides
        self.dcn_on_last_conv = dcn_on_last_conv
        assert conv_bias == 'auto' or isinstance(conv_bias, bool)
        self.conv_bias = conv_bias
        self.use_direction_classifier, self.diff_rad_by_sin = use_direction_classifier, diff_rad_by_sin
        self.dir_offset, self.dir_limit_offset = dir_offset, dir_limit_offset
        self.loss_cls, self.loss_bbox, self.loss_dir = build_loss(loss_cls), build_loss(loss_bbox), build_loss(loss_dir)
        self.bbox_code_size = bbox_code_size
        self.group_reg_dims = list(group_reg_dims)
        self.cls_branch, self.reg_branch = cls_branch, reg_branch
        assert len(reg_branch) == len(group_reg_dims)
        self.pred_velo, self.pred_bbox2d = pred_velo, pred_bbox2d
        self.out_channels = [r[-1] if len(r) > 0 else -1 for r in reg_branch]
        self.dir_branch = dir_branch
        self.train_cfg, self.test_cfg, self.conv_cfg, self.norm_cfg = train_cfg, test_cfg, conv_cfg, norm_cfg
        self.fp16_enabled = False
        self.background_label = num_classes if background_label is None else background_label
        assert self.background_label in (0, num_classes)
        self.pred_attrs, self.attr_background_label, self.num_attrs = pred_attrs, -1, num_attrs
        if self.pred_attrs:
            self.attr_background_label = num_attrs
            self.loss_attr = build_loss(loss_attr)
            self.attr_branch = attr_branch
        self._init_layers()

    # ---- layers
    def _init_layers(self):
        self.cls_convs = self._tower()
        self.reg_convs = self._tower()
        self._init_predictor()

    def _tower(self):
        convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            conv_cfg = dict(type='DCNv2') if self.dcn_on_last_conv and i == self.stacked_convs - 1 else self.conv_cfg
            convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1, conv_cfg=conv_cfg, norm_cfg=self.norm_cfg,
                                    bias=self.conv_bias))
        return convs

    def _init_branch(self, conv_channels=(64), conv_strides=(1)):
        if isinstance(conv_channels, int):
            conv_channels, conv_strides = [self.feat_channels, conv_channels], [conv_strides]
        else:
            conv_channels, conv_strides = [self.feat_channels] + list(conv_channels), list(conv_strides)
        return nn.ModuleList([ConvModule(conv_channels[i], conv_channels[i + 1], 3, stride=conv_strides[i], padding=1,
                                         conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg, bias=self.conv_bias)
                              for i in range(len(conv_strides))])

    def _init_predictor(self):
        self.conv_cls_prev = self._init_branch(self.cls_branch, (1, ) * len(self.cls_branch))
        self.conv_cls = nn.Conv2d(self.cls_branch[-1], self.cls_out_channels, 1)
        self.conv_reg_prevs, self.conv_regs = nn.ModuleList(), nn.ModuleList()
        for i, reg_dim in enumerate(self.group_reg_dims):
            if len(self.reg_branch[i]) > 0:
                self.conv_reg_prevs.append(self._init_branch(self.reg_branch[i], (1, ) * len(self.reg_branch[i])))
                self.conv_regs.append(nn.Conv2d(self.out_channels[i], reg_dim, 1))
            else:
                self.conv_reg_prevs.append(None)
                self.conv_regs.append(nn.Conv2d(self.feat_channels, reg_dim, 1))
        if self.use_direction_classifier:
            self.conv_dir_cls_prev = self._init_branch(self.dir_branch, (1, ) * len(self.dir_branch))
            self.conv_dir_cls = nn.Conv2d(self.dir_branch[-1], 2, 1)
        if self.pred_attrs:
            self.conv_attr_prev = self._init_branch(self.attr_branch, (1, ) * len(self.attr_branch))
            self.conv_attr = nn.Conv2d(self.attr_branch[-1], self.num_attrs, 1)

    def _init_branch_weights(self, branch):
        for m in branch:
            if isinstance(m.conv, nn.Conv2d):
                normal_init(m.conv, std=0.01)

    def init_weights(self):
        for modules in [self.cls_convs, self.reg_convs, self.conv_cls_prev]:
            self._init_branch_weights(modules)
        for prev in self.conv_reg_prevs:
            if prev is not None:
                self._init_branch_weights(prev)
        if self.use_direction_classifier:
            self._init_branch_weights(self.conv_dir_cls_prev)
        if self.pred_attrs:
            self._init_branch_weights(self.conv_attr_prev)
        bias_cls = bias_init_with_prob(0.01)
        normal_init(self.conv_cls, std=0.01, bias=bias_cls)
        for conv_reg in self.conv_regs:
            normal_init(conv_reg, std=0.01)
        if self.use_direction_classifier:
            normal_init(self.conv_dir_cls, std=0.01, bias=bias_cls)
        if self.pred_attrs:
            normal_init(self.conv_attr, std=0.01, bias=bias_cls)

    @staticmethod
    def _run(branch, x):
        for layer in branch:
            x = layer(x)
        return x

    def _forward_base(self, x):
        cls_feat = self._run(self.cls_convs, x)
        cls_score = self.conv_cls(self._run(self.conv_cls_prev, cls_feat))
        reg_feat = self._run(self.reg_convs, x)
        bbox_pred = []
        for i in range(len(self.group_reg_dims)):
            f = reg_feat if len(self.reg_branch[i]) == 0 else self._run(self.conv_reg_prevs[i], reg_feat)
            bbox_pred.append(self.conv_regs[i](f))
        bbox_pred = torch.cat(bbox_pred, dim=1)
        dir_cls_pred = self.conv_dir_cls(self._run(self.conv_dir_cls_prev, reg_feat)) if self.use_direction_classifier else None
        attr_pred = self.conv_attr(self._run(self.conv_attr_prev, cls_feat)) if self.pred_attrs else None
        return cls_score, bbox_pred, dir_cls_pred, attr_pred, cls_feat, reg_feat

    def _get_points_single(self, featmap_size, stride, dtype, device, flatten=False):
        h, w = featmap_size
        y, x = torch.meshgrid(torch.arange(h, dtype=dtype, device=device), torch.arange(w, dtype=dtype, device=device),
                              indexing='ij')
        return (y.flatten(), x.flatten()) if flatten else (y, x)

    def get_points(self, featmap_sizes, dtype, device, flatten=False):
        return [self._get_points_single(featmap_sizes[i], self.strides[i], dtype, device, flatten)
                for i in range(len(featmap_sizes))]


class FCOSMono3DHead(AnchorFreeMono3DHead):
    def __init__(self, regress_ranges=((-1, 48), (48, 96), (96, 192), (192, 384), (384, INF)), center_sampling=True,
                 center_sample_radius=1.5, norm_on_bbox=True, centerness_on_reg=True, centerness_alpha=2.5,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_dir=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_attr=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 bbox_coder=dict(type='FCOS3DBBoxCoder', code_size=9), norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 centerness_branch=(64, ), init_cfg=None, **kwargs):
        self.regress_ranges, self.center_sampling, self.center_sample_radius = regress_ranges, center_sampling, center_sample_radius
        self.norm_on_bbox, self.centerness_on_reg, self.centerness_alpha = norm_on_bbox, centerness_on_reg, centerness_alpha
        self.centerness_branch = centerness_branch
        super().__init__(loss_cls=loss_cls, loss_bbox=loss_bbox, loss_dir=loss_dir, loss_attr=loss_attr, norm_cfg=norm_cfg,
                         init_cfg=init_cfg, **kwargs)
        self.loss_centerness = build_loss(loss_centerness)
        bbox_coder = dict(bbox_coder)
        bbox_coder['code_size'] = self.bbox_code_size
        self.bbox_coder = build_bbox_coder(bbox_coder)

    def _init_layers(self):
        super()._init_layers()
        self.conv_centerness_prev = self._init_branch(self.centerness_branch, (1, ) * len(self.centerness_branch))
        self.conv_centerness = nn.Conv2d(self.centerness_branch[-1], 1, 1)
        self.scale_dim = 3       # offset, depth, size
        self._make_scales()

    def _make_scales(self):
        self.scales = nn.ModuleList([nn.ModuleList([Scale(1.0) for _ in range(self.scale_dim)]) for _ in self.strides])

    def init_weights(self):
        super().init_weights()
        self._init_branch_weights(self.conv_centerness_prev)
        normal_init(self.conv_centerness, std=0.01)

    def _forward_fcos(self, x, scale, stride):
        cls_score, bbox_pred, dir_cls_pred, attr_pred, cls_feat, reg_feat = self._forward_base(x)
        centerness = self.conv_centerness(self._run(self.conv_centerness_prev, reg_feat if self.centerness_on_reg else cls_feat))
        bbox_pred = self.bbox_coder.decode(bbox_pred, scale, stride, self.training, cls_score)
        return cls_score, bbox_pred, dir_cls_pred, attr_pred, centerness, cls_feat, reg_feat

    @staticmethod
    def add_sin_difference(boxes1, boxes2):
        rad_pred = torch.sin(boxes1[..., 6:7]) * torch.cos(boxes2[..., 6:7])
        rad_tg = torch.cos(boxes1[..., 6:7]) * torch.sin(boxes2[..., 6:7])
        return (torch.cat([boxes1[..., :6], rad_pred, boxes1[..., 7:]], dim=-1),
                torch.cat([boxes2[..., :6], rad_tg, boxes2[..., 7:]], dim=-1))

    @staticmethod
    def get_direction_target(reg_targets, dir_offset=0, dir_limit_offset=0.0, num_bins=2, one_hot=True):
        offset_rot = limit_period(reg_targets[..., 6] - dir_offset, dir_limit_offset, 2 * np.pi)
        t = torch.clamp(torch.floor(offset_rot / (2 * np.pi / num_bins)).long(), min=0, max=num_bins - 1)
        if one_hot:
            oh = torch.zeros(*list(t.shape), num_bins, dtype=reg_targets.dtype, device=t.device)
            oh.scatter_(t.unsqueeze(dim=-1).long(), 1.0)
            return oh
        return t

    def _get_points_single(self, featmap_size, stride, dtype, device, flatten=False):
        y, x = super()._get_points_single(featmap_size, stride, dtype, device)
        return torch.stack((x.reshape(-1) * stride, y.reshape(-1) * stride), dim=-1) + stride // 2

    # ---- targets: the whole batch in one launch
    def assign_targets(self, points, gt_bboxes_list, gt_labels_list, gt_bboxes_3d_list, gt_labels_3d_list, centers2d_list,
                       depths_list, attr_labels_list):
        """``multi_apply(self._get_target_single, ...)`` of the reference for all images at once ->
        (labels [B,P], bbox_targets [B,P,4], labels_3d [B,P], bbox_targets_3d [B,P,code], centerness [B,P], attr [B,P])."""
        assert self.center_sampling is True, 'Setting center_sampling to False has not been implemented for FCOS3D.'
        assert len(points) == len(self.regress_ranges)
        dev = points[0].device
        concat_points = torch.cat(points, dim=0).float().contiguous()
        F._need_cuda(concat_points)
        P, B, code = concat_points.shape[0], len(gt_labels_list), self.bbox_code_size
        begins = np.cumsum([0] + [p.size(0) for p in points])[:-1].astype(np.int32)
        g3d = []
        for t in gt_bboxes_3d_list:
            t = (t if isinstance(t, torch.Tensor) else t.tensor).to(dev).float().clone()
            if t.shape[0]:      # global yaw -> local yaw (the reference edits the caller's tensor in place; a copy here)
                t[..., 6] = -torch.atan2(t[..., 0], t[..., 2]) + t[..., 6]
            g3d.append(t.reshape(-1, code))
        if attr_labels_list is None:
            attr_labels_list = [g.new_full(g.shape, self.attr_background_label) for g in gt_labels_list]
        offs = torch.tensor(np.cumsum([0] + [int(g.shape[0]) for g in gt_labels_list]), dtype=torch.int64)
        cat = lambda xs, dt, w: torch.cat([x.to(dev).to(dt).reshape(-1, w) for x in xs]).contiguous() if xs else None
        gb, c2d = cat(gt_bboxes_list, torch.float32, 4), cat(centers2d_list, torch.float32, 2)
        dep, g3 = cat(depths_list, torch.float32, 1), torch.cat(g3d).contiguous()
        gl, gl3, al = cat(gt_labels_list, torch.int64, 1), cat(gt_labels_3d_list, torch.int64, 1), cat(attr_labels_list, torch.int64, 1)
        offs_d = F.upload(offs, dev)
        out_l = torch.empty((B, P), dtype=torch.int64, device=dev)
        out_l3, out_a = torch.empty_like(out_l), torch.empty_like(out_l)
        out_bt = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
        out_t3 = torch.empty((B, P, code), dtype=torch.float32, device=dev)
        out_c = torch.empty((B, P), dtype=torch.float32, device=dev)
        import ctypes as C
        strides = (C.c_float * len(self.strides))(*[float(s) for s in self.strides])
        ranges = (C.c_float * (2 * len(self.regress_ranges)))(*[float(v) for r in self.regress_ranges for v in r])
        check(_lib.lib().gga_fcos3d_targets(F._p(concat_points), P, len(points), begins.ctypes.data_as(C.POINTER(C.c_int32)), strides, ranges,
                                            float(self.center_sample_radius), F._p(offs_d), B, F._p(gb), F._p(c2d), F._p(dep), F._p(g3),
                                            code, F._p(gl), F._p(gl3), F._p(al), int(self.background_label),
                                            int(self.attr_background_label), float(self.centerness_alpha), F._p(out_l), F._p(out_bt),
                                            F._p(out_l3), F._p(out_t3), F._p(out_c), F._p(out_a), F._stream()), 'gga_fcos3d_targets')
        return out_l, out_bt, out_l3, out_t3, out_c, out_a


@HEADS.register_module()
class PGDHead(FCOSMono3DHead):
    def __init__(self, use_depth_classifier=True, use_onlyreg_proj=False, weight_dim=-1, weight_branch=((256, ), ),
                 depth_branch=(64, ), depth_range=(0, 70), depth_unit=10, division='uniform', depth_bins=8,
                 loss_depth=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_bbox2d=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_consistency=dict(type='GIoULoss', loss_weight=1.0), pred_bbox2d=True, pred_keypoints=False,
                 bbox_coder=dict(type='PGDBBoxCoder', base_depths=((28.01, 16.32), ),
                                 base_dims=((0.8, 1.73, 0.6), (1.76, 1.73, 0.6), (3.9, 1.56, 1.6)), code_size=7), **kwargs):
        self.use_depth_classifier, self.use_onlyreg_proj = use_depth_classifier, use_onlyreg_proj
        self.depth_branch, self.pred_keypoints = depth_branch, pred_keypoints
        self.weight_dim, self.weight_branch = weight_dim, weight_branch
        self.weight_out_channels = [w[-1] if len(w) > 0 else -1 for w in weight_branch]
        self.depth_range, self.depth_unit, self.division = depth_range, depth_unit, division
        if division == 'uniform':
            self.num_depth_cls = int((depth_range[1] - depth_range[0]) / depth_unit) + 1
        else:
            self.num_depth_cls = depth_bins
        super().__init__(pred_bbox2d=pred_bbox2d, bbox_coder=bbox_coder, **kwargs)
        self.loss_depth = build_loss(loss_depth)
        if self.pred_bbox2d:
            self.loss_bbox2d = build_loss(loss_bbox2d)
            self.loss_consistency = build_loss(loss_consistency)
        if self.pred_keypoints:
            self.kpts_start = 9 if self.pred_velo else 7

    def _init_layers(self):
        super()._init_layers()
        self.scale_dim += int(self.pred_bbox2d) + int(self.pred_keypoints)
        self._make_scales()

    def _init_predictor(self):
        super()._init_predictor()
        if self.use_depth_classifier:
            self.conv_depth_cls_prev = self._init_branch(self.depth_branch, (1, ) * len(self.depth_branch))
            self.conv_depth_cls = nn.Conv2d(self.depth_branch[-1], self.num_depth_cls, 1)
            self.fuse_lambda = nn.Parameter(torch.tensor(10e-5))      # "learnable weight of the depth fusion"
        if self.weight_dim != -1:
            self.conv_weight_prevs, self.conv_weights = nn.ModuleList(), nn.ModuleList()
            for i in range(self.weight_dim):
                if len(self.weight_branch[i]) > 0:
                    self.conv_weight_prevs.append(self._init_branch(self.weight_branch[i], (1, ) * len(self.weight_branch[i])))
                    self.conv_weights.append(nn.Conv2d(self.weight_out_channels[i], 1, 1))
                else:
                    self.conv_weight_prevs.append(None)
                    self.conv_weights.append(nn.Conv2d(self.feat_channels, 1, 1))

    def init_weights(self):
        super().init_weights()
        bias_cls = bias_init_with_prob(0.01)
        if self.use_depth_classifier:
            self._init_branch_weights(self.conv_depth_cls_prev)
            normal_init(self.conv_depth_cls, std=0.01, bias=bias_cls)
        if self.weight_dim != -1:
            for prev in self.conv_weight_prevs:
                if prev is not None:
                    self._init_branch_weights(prev)
            for conv_weight in self.conv_weights:
                normal_init(conv_weight, std=0.01)

    def forward(self, feats):
        if LEVEL_BATCH and len(feats) > 1 and feats[0].is_cuda:
            return self.forward_levels(feats)
        if not (LEVEL_STREAMS and len(feats) > 1 and feats[0].is_cuda):
            return multi_apply(self.forward_single, feats, self.scales, self.strides)
        # The levels are independent until the loss, and from the second one on their maps are too small to fill the chip
        # (a 24 x 78 map is 108 tiles of the convolution kernel, the 3 x 10 one 12): every level runs on a stream of its own,
        # so the device overlaps their kernels - forward here, and backward too (autograd runs a node's backward on the
        # stream of its forward and orders the streams itself).
        main = torch.cuda.current_stream(feats[0].device)
        streams = self._level_streams(len(feats), feats[0].device, main)
        # The levels share every parameter. Each level reads them through an alias made HERE, on the main stream: in the
        # backward pass the alias node - which runs on the stream of its forward, this one - is where the levels' gradients
        # meet, so a parameter's gradient is accumulated on the main stream like everywhere else in the model (and under
        # DistributedDataParallel the reducer's hooks see one stream). Without it autograd accumulates on whichever level
        # stream delivers first and says so ("AccumulateGrad node's stream does not match ...").
        # Every level gets aliases of its OWN (_ShareParams): the five gradients of a parameter then arrive as five inputs of one
        # backward node, which adds them with four multi-tensor launches for ALL parameters - one shared alias made autograd add
        # them one by one as they came in (4 additions per parameter, ~400 of the step's 626 small add kernels: 9 ms of 235).
        from torch.nn.utils import stateless
        named = [(k, p) for k, p in self.named_parameters() if p.requires_grad] if torch.is_grad_enabled() else []
        if named and os.environ.get('GGA_PGD_SHARED_ALIAS') == '1':       # (A/B: the one alias per parameter of round 3's first form)
            one = [p.view_as(p) for _, p in named]
            copies = tuple(one) * len(feats)
        else:
            copies = _ShareParams.apply(len(feats), *[p for _, p in named]) if named else ()
        outs = []
        for li, (x, scale, stride, st) in enumerate(zip(feats, self.scales, self.strides, streams)):
            if st is not main:
                st.wait_stream(main)
                x.record_stream(st)
            alias = {k: copies[li * len(named) + i] for i, (k, _) in enumerate(named)}
            with torch.cuda.stream(st), stateless._reparametrize_module(self, alias):
                o = self.forward_single(x, scale, stride)
            if st is not main:
                for t in o:
                    if isinstance(t, torch.Tensor):
                        t.record_stream(main)      # produced on the level's stream, read by the loss on this one
            outs.append(o)
        for st in streams:
            if st is not main:
                main.wait_stream(st)
        return tuple(map(list, zip(*outs)))

    @staticmethod
    def _run_lv(branch, xs):
        for layer in branch:
            xs = layer.forward_levels(xs) if hasattr(layer, 'forward_levels') else [layer(x) for x in xs]
        return xs

    def forward_levels(self, feats):
        """``forward`` layer by layer over all levels instead of level by level over all layers: the same modules on the
        same tensors (``_forward_base`` / ``_forward_fcos`` / ``forward_single`` with lists), so that every convolution
        whose weights the levels share is ONE launch over the five maps (ConvModule.forward_levels)."""
        n, run = len(feats), self._run_lv
        each = lambda conv, xs: [conv(x) for x in xs]
        cls_feat = run(self.cls_convs, feats)
        cls_score = each(self.conv_cls, run(self.conv_cls_prev, cls_feat))
        reg_feat = run(self.reg_convs, feats)
        parts = []
        for i in range(len(self.group_reg_dims)):
            f = reg_feat if len(self.reg_branch[i]) == 0 else run(self.conv_reg_prevs[i], reg_feat)
            parts.append(each(self.conv_regs[i], f))
        bbox_pred = [torch.cat([p[l] for p in parts], dim=1) for l in range(n)]
        dir_cls_pred = each(self.conv_dir_cls, run(self.conv_dir_cls_prev, reg_feat)) if self.use_direction_classifier else [None] * n
        attr_pred = each(self.conv_attr, run(self.conv_attr_prev, cls_feat)) if self.pred_attrs else [None] * n
        centerness = each(self.conv_centerness, run(self.conv_centerness_prev, reg_feat if self.centerness_on_reg else cls_feat))
        depth_cls_pred = each(self.conv_depth_cls, run(self.conv_depth_cls_prev, reg_feat)) if self.use_depth_classifier else [None] * n
        weight = [None] * n
        if self.weight_dim != -1:
            ws = [each(self.conv_weights[i], reg_feat if len(self.weight_branch[i]) == 0 else run(self.conv_weight_prevs[i], reg_feat))
                  for i in range(self.weight_dim)]
            weight = [torch.cat([w[l] for w in ws], dim=1) for l in range(n)]
        for l, (scale, stride) in enumerate(zip(self.scales, self.strides)):
            bbox_pred[l] = self.bbox_coder.decode(bbox_pred[l], scale, stride, self.training, cls_score[l])
            max_regress_range = stride * self.regress_ranges[0][1] / self.strides[0]
            bbox_pred[l] = self.bbox_coder.decode_2d(bbox_pred[l], scale, stride, max_regress_range, self.training,
                                                     self.pred_keypoints, self.pred_bbox2d)
        return cls_score, bbox_pred, dir_cls_pred, depth_cls_pred, weight, attr_pred, centerness

    @staticmethod
    def _level_streams(n, device, main):
        key = (str(device), n)                     # per process, not per module: a module must stay copyable / picklable
        if key not in _LEVEL_STREAM_CACHE:
            _LEVEL_STREAM_CACHE[key] = [torch.cuda.Stream(device=device) for _ in range(n - 1)]
        return [main] + _LEVEL_STREAM_CACHE[key]   # the largest level stays on the caller's stream

    def forward_single(self, x, scale, stride):
        cls_score, bbox_pred, dir_cls_pred, attr_pred, centerness, cls_feat, reg_feat = self._forward_fcos(x, scale, stride)
        max_regress_range = stride * self.regress_ranges[0][1] / self.strides[0]
        bbox_pred = self.bbox_coder.decode_2d(bbox_pred, scale, stride, max_regress_range, self.training, self.pred_keypoints,
                                              self.pred_bbox2d)
        depth_cls_pred = self.conv_depth_cls(self._run(self.conv_depth_cls_prev, reg_feat)) if self.use_depth_classifier else None
        weight = None
        if self.weight_dim != -1:
            weight = torch.cat([self.conv_weights[i](reg_feat if len(self.weight_branch[i]) == 0 else
                                                     self._run(self.conv_weight_prevs[i], reg_feat))
                                for i in range(self.weight_dim)], dim=1)
        return cls_score, bbox_pred, dir_cls_pred, depth_cls_pred, weight, attr_pred, centerness

    # ---- targets per level (pgd_head.py:1132-1229)
    def get_targets(self, points, gt_bboxes_list, gt_labels_list, gt_bboxes_3d_list, gt_labels_3d_list, centers2d_list,
                    depths_list, attr_labels_list):
        num_points = [p.size(0) for p in points]
        _, bt, l3, t3, cen, attr = self.assign_targets(points, gt_bboxes_list, gt_labels_list, gt_bboxes_3d_list, gt_labels_3d_list,
                                                       centers2d_list, depths_list, attr_labels_list)
        labels_3d, targets_3d, centerness, attrs = [], [], [], []
        for i, (lo, n) in enumerate(zip(np.cumsum([0] + num_points)[:-1], num_points)):
            sl = slice(int(lo), int(lo) + n)
            labels_3d.append(l3[:, sl].reshape(-1))                       # image-major inside a level, as torch.cat over images
            centerness.append(cen[:, sl].reshape(-1))
            attrs.append(attr[:, sl].reshape(-1))
            t = t3[:, sl].reshape(-1, self.bbox_code_size)
            if self.pred_bbox2d:
                t = torch.cat([t, bt[:, sl].reshape(-1, 4)], dim=1)
            else:
                t = t.clone()
            if self.norm_on_bbox:
                t[:, :2] = t[:, :2] / self.strides[i]
                if self.pred_bbox2d:
                    t[:, -4:] = t[:, -4:] / self.strides[i]
            targets_3d.append(t)
        return labels_3d, targets_3d, centerness, attrs

    def get_pos_predictions(self, bbox_preds, dir_cls_preds, depth_cls_preds, weights, attr_preds, centernesses, pos_inds, img_metas):
        flat = lambda xs, w: torch.cat([x.permute(0, 2, 3, 1).reshape(-1, w) for x in xs])
        pos_bbox_preds = flat(bbox_preds, sum(self.group_reg_dims))[pos_inds]
        pos_dir_cls_preds = flat(dir_cls_preds, 2)[pos_inds]
        pos_centerness = torch.cat([c.permute(0, 2, 3, 1).reshape(-1) for c in centernesses])[pos_inds]
        pos_depth_cls_preds = flat(depth_cls_preds, self.num_depth_cls)[pos_inds] if self.use_depth_classifier else None
        pos_weights = flat(weights, self.weight_dim)[pos_inds] if self.weight_dim != -1 else None
        pos_attr_preds = flat(attr_preds, self.num_attrs)[pos_inds] if self.pred_attrs else None
        return pos_bbox_preds, pos_dir_cls_preds, pos_depth_cls_preds, pos_weights, pos_attr_preds, pos_centerness

    def get_proj_bbox2d(self, bbox_preds, pos_dir_cls_preds, labels_3d, bbox_targets_3d, pos_points, pos_inds, img_metas,
                        pos_depth_cls_preds=None, pos_weights=None, pos_cls_scores=None, with_kpts=False):
        """2D boxes of the projected 3D predictions, the decoded 2D predictions and (optionally) the key-point
        targets of the positive points (pgd_head.py:265-441)."""
        views = [np.array(m['cam2img']) for m in img_metas]
        num_imgs = len(img_metas)
        img_pos = self._img_pos          # per image: positions of its points in the positive set (prepare_loss), None = none
        sp, sp2d, st, ss = [], [], [], []
        for i, bbox_pred in enumerate(bbox_preds):
            f = bbox_pred.permute(0, 2, 3, 1).reshape(-1, sum(self.group_reg_dims))
            f = torch.cat([f[:, :2] * self.strides[i], f[:, 2:-4], f[:, -4:] * self.strides[i]], dim=1)
            sp.append(f[:, :self.bbox_coder.bbox_code_size])
            sp2d.append(f[:, -4:])
            t = bbox_targets_3d[i]
            st.append(torch.cat([t[:, :2] * self.strides[i], t[:, 2:-4], t[:, -4:] * self.strides[i]], dim=1))
            ss.append(f.new_ones(f.shape[0], 1) * self.strides[i])
        pos_preds = torch.cat(sp)[pos_inds]
        pos_bbox2d = torch.cat(sp2d)[pos_inds]
        pos_targets = torch.cat(st)[pos_inds]
        pos_strides = torch.cat(ss)[pos_inds]
        pos_decoded_bbox2d_preds = distance2bbox(pos_points, pos_bbox2d)
        pos_preds = torch.cat([pos_points - pos_preds[:, :2], pos_preds[:, 2:]], dim=1)
        pos_targets = torch.cat([pos_points - pos_targets[:, :2], pos_targets[:, 2:]], dim=1)
        if self.use_depth_classifier and not self.use_onlyreg_proj:
            prob = self.bbox_coder.decode_prob_depth(pos_depth_cls_preds, self.depth_range, self.depth_unit, self.division,
                                                     self.num_depth_cls)
            a = torch.sigmoid(self.fuse_lambda)
            pos_preds = torch.cat([pos_preds[:, :2], (a * pos_preds[:, 2] + (1 - a) * prob)[:, None], pos_preds[:, 3:]], dim=1)
        corners_img = pos_preds.new_zeros((*pos_preds.shape[:-1], 8, 2))
        corners_img_gt = pos_preds.new_zeros((*pos_preds.shape[:-1], 8, 2))
        box_type = img_metas[0]['box_type_3d']
        code = self.bbox_coder.bbox_code_size
        for idx in range(num_imgs):
            mask = img_pos[idx]
            if mask is None:
                continue
            view = views[idx]
            cam2img = torch.eye(4, dtype=pos_preds.dtype, device=pos_preds.device)
            cam2img[:view.shape[0], :view.shape[1]] = F.const_tensor(np.ascontiguousarray(view), pos_preds.device, pos_preds.dtype)
            p, t = pos_preds[mask], pos_targets[mask]
            centers2d_preds, centers2d_targets = p[:, :2].clone(), t[:, :2].clone()
            t3 = points_img2cam(t[:, :3], view)
            p3 = points_img2cam(p[:, :3], view)
            p = torch.cat([p3[:, :2], t3[:, 2:3], p[:, 3:]], dim=1)       # the depth of the target: only the rest is judged
            t = torch.cat([t3, t[:, 3:]], dim=1)
            if self.use_direction_classifier:
                dir_cls = torch.max(pos_dir_cls_preds[mask], dim=-1)[1]
                p = self.bbox_coder.decode_yaw(p.clone(), centers2d_preds, dir_cls, self.dir_offset, cam2img)
            t = torch.cat([t[:, :6], (torch.atan2(centers2d_targets[:, 0] - cam2img[0, 2], cam2img[0, 0]) + t[:, 6])[:, None],
                           t[:, 7:]], dim=1)
            corners_img[mask] = points_cam2img(box_type(p[:, :code], box_dim=code, origin=(0.5, 0.5, 0.5)).corners, cam2img)
            corners_img_gt[mask] = points_cam2img(box_type(t[:, :self.bbox_code_size], box_dim=code, origin=(0.5, 0.5, 0.5)).corners,
                                                  cam2img)
        proj = torch.cat([torch.min(corners_img, dim=1)[0], torch.max(corners_img, dim=1)[0]], dim=1)
        outputs = (proj, pos_decoded_bbox2d_preds)
        if with_kpts:
            norm_strides = pos_strides * self.regress_ranges[0][1] / self.strides[0]
            kpts = (corners_img_gt - pos_points[..., None, :]).view((*pos_preds.shape[:-1], 16)) / norm_strides
            outputs += (kpts, )
        return outputs

    # ---- inference (pgd_head.py:878-1130)
    @torch.no_grad()
    def get_bboxes(self, cls_scores, bbox_preds, dir_cls_preds, depth_cls_preds, weights, attr_preds, centernesses, img_metas,
                   cfg=None, rescale=None):
        num_levels = len(cls_scores)
        mlvl_points = self.get_points([f.size()[-2:] for f in cls_scores], bbox_preds[0].dtype, bbox_preds[0].device)
        result_list = []
        for img_id in range(len(img_metas)):
            pick = lambda xs: [xs[i][img_id].detach() for i in range(num_levels)]
            cls_list = pick(cls_scores)
            full = lambda c, v=0: [cls_list[i].new_full([c, *cls_list[i].shape[1:]], v) for i in range(num_levels)]
            result_list.append(self._get_bboxes_single(
                cls_list, pick(bbox_preds), pick(dir_cls_preds) if self.use_direction_classifier else full(2),
                pick(depth_cls_preds) if self.use_depth_classifier else full(self.num_depth_cls),
                pick(weights) if self.weight_dim != -1 else full(1),
                pick(attr_preds) if self.pred_attrs else full(self.num_attrs, self.attr_background_label),
                pick(centernesses), mlvl_points, img_metas[img_id], cfg, rescale))
        return result_list

    def _get_bboxes_single(self, cls_scores, bbox_preds, dir_cls_preds, depth_cls_preds, weights, attr_preds, centernesses,
                           mlvl_points, input_meta, cfg, rescale=False):
        from . import ops
        view = np.array(input_meta['cam2img'])
        scale_factor = input_meta['scale_factor']
        cfg = self.test_cfg if cfg is None else cfg
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        code = self.bbox_coder.bbox_code_size
        acc = dict(c2d=[], box=[], score=[], dir=[], attr=[], cen=[], dcls=[], dunc=[], b2d=[])
        for cls_score, bbox_pred, dir_cls_pred, depth_cls_pred, weight, attr_pred, centerness, points in zip(
                cls_scores, bbox_preds, dir_cls_preds, depth_cls_preds, weights, attr_preds, centernesses, mlvl_points):
            scores = cls_score.permute(1, 2, 0).reshape(-1, self.cls_out_channels).sigmoid()
            dir_cls_pred = dir_cls_pred.permute(1, 2, 0).reshape(-1, 2)
            dir_cls_score = torch.max(dir_cls_pred, dim=-1)[1]
            depth_cls_pred = depth_cls_pred.permute(1, 2, 0).reshape(-1, self.num_depth_cls)
            depth_cls_score = torch.softmax(depth_cls_pred, dim=-1).topk(k=2, dim=-1)[0].mean(dim=-1)
            weight = weight.permute(1, 2, 0).reshape(-1, self.weight_dim if self.weight_dim != -1 else 1)
            depth_uncertainty = torch.exp(-weight[:, -1])
            attr_score = torch.max(attr_pred.permute(1, 2, 0).reshape(-1, self.num_attrs), dim=-1)[1]
            centerness = centerness.permute(1, 2, 0).reshape(-1).sigmoid()
            bbox_pred = bbox_pred.permute(1, 2, 0).reshape(-1, sum(self.group_reg_dims))
            bbox_pred3d = bbox_pred[:, :code].clone()
            bbox_pred2d = bbox_pred[:, -4:].clone() if self.pred_bbox2d else None
            nms_pre = get('nms_pre', -1)
            if nms_pre > 0 and scores.shape[0] > nms_pre:
                merged = scores * centerness[:, None]
                if self.use_depth_classifier:
                    merged = merged * depth_cls_score[:, None]
                    if self.weight_dim != -1:
                        merged = merged * depth_uncertainty[:, None]
                topk_inds = merged.max(dim=1)[0].topk(nms_pre)[1]
                points, bbox_pred3d, scores = points[topk_inds, :], bbox_pred3d[topk_inds, :], scores[topk_inds, :]
                depth_cls_pred, centerness, dir_cls_score = depth_cls_pred[topk_inds, :], centerness[topk_inds], dir_cls_score[topk_inds]
                depth_cls_score, depth_uncertainty, attr_score = depth_cls_score[topk_inds], depth_uncertainty[topk_inds], attr_score[topk_inds]
                if self.pred_bbox2d:
                    bbox_pred2d = bbox_pred2d[topk_inds, :]
            bbox_pred3d[:, :2] = points - bbox_pred3d[:, :2]
            if rescale:
                bbox_pred3d[:, :2] /= bbox_pred3d[:, :2].new_tensor(scale_factor)
                if self.pred_bbox2d:
                    bbox_pred2d /= bbox_pred2d.new_tensor(scale_factor)
            if self.use_depth_classifier:
                prob = self.bbox_coder.decode_prob_depth(depth_cls_pred, self.depth_range, self.depth_unit, self.division,
                                                         self.num_depth_cls)
                a = torch.sigmoid(self.fuse_lambda)
                bbox_pred3d[:, 2] = a * bbox_pred3d[:, 2] + (1 - a) * prob
            acc['c2d'].append(bbox_pred3d[:, :3].clone())
            bbox_pred3d[:, :3] = points_img2cam(bbox_pred3d[:, :3], view)
            acc['box'].append(bbox_pred3d), acc['score'].append(scores), acc['dir'].append(dir_cls_score)
            acc['dcls'].append(depth_cls_score), acc['attr'].append(attr_score), acc['cen'].append(centerness)
            acc['dunc'].append(depth_uncertainty)
            if self.pred_bbox2d:
                acc['b2d'].append(distance2bbox(points, bbox_pred2d, max_shape=input_meta['img_shape']))
        centers2d, bboxes, dir_scores = torch.cat(acc['c2d']), torch.cat(acc['box']), torch.cat(acc['dir'])
        bboxes2d = torch.cat(acc['b2d']) if self.pred_bbox2d else None
        cam2img = torch.eye(4, dtype=centers2d.dtype, device=centers2d.device)
        cam2img[:view.shape[0], :view.shape[1]] = F.const_tensor(np.ascontiguousarray(view), centers2d.device, centers2d.dtype)
        bboxes = self.bbox_coder.decode_yaw(bboxes, centers2d, dir_scores, self.dir_offset, cam2img)
        box_type = input_meta['box_type_3d']
        for_nms = ops.xywhr2xyxyr(box_type(bboxes, box_dim=code, origin=(0.5, 0.5, 0.5)).bev)
        scores = torch.cat(acc['score'])
        scores = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], dim=1)
        nms_scores = scores * torch.cat(acc['cen'])[:, None]
        if self.use_depth_classifier:
            nms_scores = nms_scores * torch.cat(acc['dcls'])[:, None]
            if self.weight_dim != -1:
                nms_scores = nms_scores * torch.cat(acc['dunc'])[:, None]
        results = ops.box3d_multiclass_nms(bboxes, for_nms, nms_scores, get('score_thr'), get('max_per_img'), cfg, dir_scores,
                                           torch.cat(acc['attr']), bboxes2d)
        out_boxes, out_scores, labels, _, attrs = results[0:5]
        out_boxes = box_type(out_boxes, box_dim=code, origin=(0.5, 0.5, 0.5))
        outputs = (out_boxes, out_scores, labels, attrs.to(labels.dtype) if self.pred_attrs else None)
        if self.pred_bbox2d:
            outputs = outputs + (torch.cat([results[-1], out_scores[:, None]], dim=1), )
        return outputs

    prepare_loss_before_forward = True      # SingleStageMono3DDetector.forward_train calls prepare_loss before the forward pass

    def prepare_loss(self, featmap_sizes, num_imgs, dtype, device, gt_bboxes, gt_labels, gt_bboxes_3d, gt_labels_3d, centers2d,
                     depths, attr_labels):
        """Everything of ``loss`` that depends on the ground truth only: the per-level targets (pgd_head.py:1132-1229), the
        indices of the positive points and their split by image. The sizes of those index sets are data dependent, i.e.
        the host waits for the device here - which is why the detector calls this BEFORE it queues the forward pass
        (the wait then ends with the previous step's kernels, and the host work of the loss section runs while the
        device is busy with the forward pass) instead of right after it (the device then idles for that host work)."""
        points = self.get_points(featmap_sizes, dtype, device)
        labels_3d, bbox_targets_3d, centerness_targets, attr_targets = self.get_targets(
            points, gt_bboxes, gt_labels, gt_bboxes_3d, gt_labels_3d, centers2d, depths, attr_labels)
        flatten_labels_3d = torch.cat(labels_3d)
        pos_inds = ((flatten_labels_3d >= 0) & (flatten_labels_3d < self.num_classes)).nonzero().reshape(-1)
        num_pos = len(pos_inds)
        # positions (in the positive set) of every image's points, ascending - what a boolean mask per image selects
        img_idx = torch.cat([labels_3d[0].new_ones(int(len(label) / num_imgs)) * idx for label in labels_3d for idx in range(num_imgs)])
        pos_img_idx = img_idx[pos_inds]
        counts = torch.bincount(pos_img_idx, minlength=num_imgs).tolist() if num_pos else [0] * num_imgs
        order = torch.argsort(pos_img_idx, stable=True)
        img_pos, lo = [], 0
        for n in counts:
            img_pos.append(order[lo:lo + n] if n else None)
            lo += n
        return dict(featmap_sizes=[tuple(int(v) for v in f) for f in featmap_sizes], num_imgs=num_imgs, points=points,
                    labels_3d=labels_3d, bbox_targets_3d=bbox_targets_3d, centerness_targets=centerness_targets,
                    attr_targets=attr_targets, flatten_labels_3d=flatten_labels_3d, pos_inds=pos_inds, num_pos=num_pos,
                    img_pos=img_pos)

    def featmap_sizes_of(self, img_shape):
        """Sizes of the FPN levels for an input of ``img_shape`` (.., H, W): every stride-2 stage of the backbone and the
        extra FPN convolutions (3x3 / 7x7 with 'same'-style padding, 3x3 max-pool with padding 1) halve rounding up."""
        sizes = []
        for st in self.strides:
            h, w = int(img_shape[-2]), int(img_shape[-1])
            for _ in range(int(round(math.log2(st)))):
                h, w = (h + 1) // 2, (w + 1) // 2
            sizes.append((h, w))
        return sizes

    def loss(self, cls_scores, bbox_preds, dir_cls_preds, depth_cls_preds, weights, attr_preds, centernesses, gt_bboxes, gt_labels,
             gt_bboxes_3d, gt_labels_3d, centers2d, depths, attr_labels, img_metas, gt_bboxes_ignore=None, prepared=None):
        assert len(cls_scores) == len(bbox_preds) == len(dir_cls_preds) == len(depth_cls_preds) == len(weights) == \
            len(centernesses) == len(attr_preds)
        featmap_sizes = [tuple(int(v) for v in f.size()[-2:]) for f in cls_scores]
        num_imgs = cls_scores[0].size(0)
        if prepared is None or prepared['featmap_sizes'] != featmap_sizes or prepared['num_imgs'] != num_imgs:
            prepared = self.prepare_loss(featmap_sizes, num_imgs, bbox_preds[0].dtype, bbox_preds[0].device, gt_bboxes, gt_labels,
                                         gt_bboxes_3d, gt_labels_3d, centers2d, depths, attr_labels)
        all_level_points, labels_3d, bbox_targets_3d = prepared['points'], prepared['labels_3d'], prepared['bbox_targets_3d']
        centerness_targets, attr_targets = prepared['centerness_targets'], prepared['attr_targets']
        flatten_cls_scores = torch.cat([c.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels) for c in cls_scores])
        flatten_labels_3d = prepared['flatten_labels_3d']
        flatten_bbox_targets_3d = torch.cat(bbox_targets_3d)
        flatten_centerness_targets = torch.cat(centerness_targets)
        flatten_points = torch.cat([p.repeat(num_imgs, 1) for p in all_level_points])
        pos_inds, num_pos = prepared['pos_inds'], prepared['num_pos']
        self._img_pos = prepared['img_pos']
        loss_dict = dict()
        loss_dict['loss_cls'] = self.loss_cls(flatten_cls_scores, flatten_labels_3d, avg_factor=num_pos + num_imgs)
        pos_bbox_preds, pos_dir_cls_preds, pos_depth_cls_preds, pos_weights, pos_attr_preds, pos_centerness = \
            self.get_pos_predictions(bbox_preds, dir_cls_preds, depth_cls_preds, weights, attr_preds, centernesses, pos_inds, img_metas)
        if num_pos > 0:
            pos_bbox_targets_3d = flatten_bbox_targets_3d[pos_inds]
            pos_centerness_targets = flatten_centerness_targets[pos_inds]
            pos_points = flatten_points[pos_inds]
            if self.pred_attrs:
                pos_attr_targets = torch.cat(attr_targets)[pos_inds]
            if self.use_direction_classifier:
                pos_dir_cls_targets = self.get_direction_target(pos_bbox_targets_3d, self.dir_offset, one_hot=False)
            bbox_weights = pos_centerness_targets.new_ones(len(pos_centerness_targets), sum(self.group_reg_dims))
            equal_weights = pos_centerness_targets.new_ones(pos_centerness_targets.shape)
            code_weight = self.train_cfg.get('code_weight', None)
            if code_weight:
                assert len(code_weight) == sum(self.group_reg_dims)
                bbox_weights = bbox_weights * F.const_tensor(list(code_weight), bbox_weights.device, bbox_weights.dtype)
            if self.diff_rad_by_sin:
                pos_bbox_preds, pos_bbox_targets_3d = self.add_sin_difference(pos_bbox_preds, pos_bbox_targets_3d)
            avg = equal_weights.sum()
            lb = lambda a, b: self.loss_bbox(pos_bbox_preds[:, a:b], pos_bbox_targets_3d[:, a:b], weight=bbox_weights[:, a:b], avg_factor=avg)
            loss_dict['loss_offset'] = lb(0, 2)
            loss_dict['loss_size'] = lb(3, 6)
            loss_dict['loss_rotsin'] = self.loss_bbox(pos_bbox_preds[:, 6], pos_bbox_targets_3d[:, 6], weight=bbox_weights[:, 6],
                                                      avg_factor=avg)
            if self.pred_velo:
                loss_dict['loss_velo'] = lb(7, 9)
            proj_inputs = (bbox_preds, pos_dir_cls_preds, labels_3d, bbox_targets_3d, pos_points, pos_inds, img_metas)
            if self.use_direction_classifier:
                loss_dict['loss_dir'] = self.loss_dir(pos_dir_cls_preds, pos_dir_cls_targets, equal_weights, avg_factor=avg)
            loss_dict['loss_depth'] = self.loss_bbox(pos_bbox_preds[:, 2], pos_bbox_targets_3d[:, 2], weight=bbox_weights[:, 2],
                                                     avg_factor=avg)
            if self.use_depth_classifier:
                prob = self.bbox_coder.decode_prob_depth(pos_depth_cls_preds, self.depth_range, self.depth_unit, self.division,
                                                         self.num_depth_cls)
                a = torch.sigmoid(self.fuse_lambda)
                fused = a * pos_bbox_preds[:, 2] + (1 - a) * prob
                if self.weight_dim != -1:
                    loss_dict['loss_depth'] = self.loss_depth(fused, pos_bbox_targets_3d[:, 2], sigma=pos_weights[:, 0],
                                                              weight=bbox_weights[:, 2], avg_factor=avg)
                else:
                    loss_dict['loss_depth'] = self.loss_depth(fused, pos_bbox_targets_3d[:, 2], weight=bbox_weights[:, 2], avg_factor=avg)
                proj_inputs += (pos_depth_cls_preds, )
            if self.pred_keypoints:
                proj_bbox2d_preds, pos_decoded_bbox2d_preds, kpts_targets = self.get_proj_bbox2d(*proj_inputs, with_kpts=True)
                ks = self.kpts_start
                loss_dict['loss_kpts'] = self.loss_bbox(pos_bbox_preds[:, ks:ks + 16], kpts_targets, weight=bbox_weights[:, ks:ks + 16],
                                                        avg_factor=avg)
            if self.pred_bbox2d:
                loss_dict['loss_bbox2d'] = self.loss_bbox2d(pos_bbox_preds[:, -4:], pos_bbox_targets_3d[:, -4:],
                                                            weight=bbox_weights[:, -4:], avg_factor=avg)
                if not self.pred_keypoints:
                    proj_bbox2d_preds, pos_decoded_bbox2d_preds = self.get_proj_bbox2d(*proj_inputs)
                loss_dict['loss_consistency'] = self.loss_consistency(proj_bbox2d_preds, pos_decoded_bbox2d_preds,
                                                                      weight=bbox_weights[:, -4:], avg_factor=avg)
            loss_dict['loss_centerness'] = self.loss_centerness(pos_centerness, pos_centerness_targets)
            if self.pred_attrs:
                loss_dict['loss_attr'] = self.loss_attr(pos_attr_preds, pos_attr_targets, pos_centerness_targets,
                                                        avg_factor=pos_centerness_targets.sum())
        else:       # no positive point: every branch still takes part in the graph
            loss_dict['loss_offset'] = pos_bbox_preds[:, :2].sum()
            loss_dict['loss_size'] = pos_bbox_preds[:, 3:6].sum()
            loss_dict['loss_rotsin'] = pos_bbox_preds[:, 6].sum()
            loss_dict['loss_depth'] = pos_bbox_preds[:, 2].sum()
            if self.pred_velo:
                loss_dict['loss_velo'] = pos_bbox_preds[:, 7:9].sum()
            if self.pred_keypoints:
                loss_dict['loss_kpts'] = pos_bbox_preds[:, self.kpts_start:self.kpts_start + 16].sum()
            if self.pred_bbox2d:
                loss_dict['loss_bbox2d'] = pos_bbox_preds[:, -4:].sum()
                loss_dict['loss_consistency'] = pos_bbox_preds[:, -4:].sum()
            loss_dict['loss_centerness'] = pos_centerness.sum()
            if self.use_direction_classifier:
                loss_dict['loss_dir'] = pos_dir_cls_preds.sum()
            if self.use_depth_classifier:
                a = torch.sigmoid(self.fuse_lambda)
                fuse = a * pos_bbox_preds[:, 2].sum() + (1 - a) * pos_depth_cls_preds.sum()
                if self.weight_dim != -1:
                    fuse = fuse * torch.exp(-pos_weights[:, 0].sum())
                loss_dict['loss_depth'] = fuse
            if self.pred_attrs:
                loss_dict['loss_attr'] = pos_attr_preds.sum()
        return loss_dict
