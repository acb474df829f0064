"""``MVXTwoStageDetector_GGA`` and ``GGA`` detectors — registry names, constructor keys and
``forward_train`` signature of the reference (mmdet3d/models/detectors/
mvx_two_stage_gga.py:20-295, centerpoint_gga.py:10-86), plus the ``train_step`` /
``_parse_losses`` the reference inherits from mmdet's ``BaseDetector`` (third-party;
restated: the total is the sum of every entry whose key contains ``'loss'``).

LiDAR-only: the GGA configs build no image branch (extract_img_feat returns None).
"""
from collections import OrderedDict

import torch
from torch import nn

from .registry import (DETECTORS, build_backbone, build_head, build_middle_encoder, build_neck,
                       build_voxel_encoder)
from .voxel_layer import Voxelization


def _sub(cfg, key):
    if cfg is None:
        return None
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class PreparedInputs:
    """The point-only front of a step, computed ahead of it (``prepare_inputs``): voxels,
    per-voxel counts, coordinates (carrying the sparse encoder's index plan when there is one)."""

    def __init__(self, voxels, num_points, coors, n_frames):
        self.voxels, self.num_points, self.coors, self.n_frames = voxels, num_points, coors, n_frames

    def __len__(self):
        return self.n_frames

    def tensors(self):
        yield from (self.voxels, self.num_points, self.coors)
        for extra in (getattr(self.coors, 'num_valid', None),):
            if extra is not None:
                yield extra
        plan = getattr(self.coors, 'index_plan', None)
        if plan is not None:
            yield from plan.tensors()


@DETECTORS.register_module()
class MVXTwoStageDetector_GGA(nn.Module):
    def __init__(self, pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None,
                 pts_fusion_layer=None, img_backbone=None, pts_backbone=None, img_neck=None, pts_neck=None,
                 pts_bbox_head=None, img_roi_head=None, img_rpn_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__()
        for name, val in (('pts_fusion_layer', pts_fusion_layer), ('img_backbone', img_backbone),
                          ('img_neck', img_neck), ('img_roi_head', img_roi_head), ('img_rpn_head', img_rpn_head)):
            if val:
                raise NotImplementedError(f'{name}: the GGA configs are LiDAR-only; the image branch is out of scope')
        self.fp16_enabled = False
        if pts_voxel_layer:
            self.pts_voxel_layer = Voxelization(**pts_voxel_layer)
        if pts_voxel_encoder:
            self.pts_voxel_encoder = build_voxel_encoder(pts_voxel_encoder)
        if pts_middle_encoder:
            self.pts_middle_encoder = build_middle_encoder(pts_middle_encoder)
        if pts_backbone:
            self.pts_backbone = build_backbone(pts_backbone)
        if pts_neck is not None:
            self.pts_neck = build_neck(pts_neck)
        if pts_bbox_head:
            pts_bbox_head = dict(pts_bbox_head)
            pts_bbox_head.update(train_cfg=_sub(train_cfg, 'pts') if train_cfg else None)
            pts_bbox_head.update(test_cfg=_sub(test_cfg, 'pts') if test_cfg else None)
            self.pts_bbox_head = build_head(pts_bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg

    with_pts_bbox = property(lambda self: getattr(self, 'pts_bbox_head', None) is not None)
    with_pts_neck = property(lambda self: getattr(self, 'pts_neck', None) is not None)
    with_pts_backbone = property(lambda self: getattr(self, 'pts_backbone', None) is not None)
    with_voxel_encoder = property(lambda self: getattr(self, 'pts_voxel_encoder', None) is not None)
    with_middle_encoder = property(lambda self: getattr(self, 'pts_middle_encoder', None) is not None)
    with_img_backbone = property(lambda self: False)

    def extract_img_feat(self, img, img_metas):
        return None

    def extract_pts_feat(self, pts, img_feats, img_metas):
        if not self.with_pts_bbox:
            return None
        if isinstance(pts, PreparedInputs):
            voxels, num_points, coors = pts.voxels, pts.num_points, pts.coors
        else:
            voxels, num_points, coors = self.voxelize(pts)
        voxel_features = self.pts_voxel_encoder(voxels, num_points, coors)
        batch_size = len(pts)      # the reference reads coors[-1, 0] + 1 back from the device
        x = self.pts_middle_encoder(voxel_features, coors, batch_size)
        x = self.pts_backbone(x)
        if self.with_pts_neck:
            x = self.pts_neck(x)
        return x

    def extract_feat(self, points, img, img_metas):
        img_feats = self.extract_img_feat(img, img_metas)
        pts_feats = self.extract_pts_feat(points, img_feats, img_metas)
        return (img_feats, pts_feats)

    @torch.no_grad()
    def voxelize(self, points):
        """list of [N_b, C] -> voxels [SM,P,C], num_points [SM], coors_batch [SM,4] (b,z,y,x):
        one batched HIP call instead of the per-frame loop + cat + pad of the reference."""
        # no host read-back when both consumers take the device-side pillar count (fused PFN + scatter):
        # the buffers then keep their capacity B * max_voxels and `coors.num_valid` carries the count
        sync = not (getattr(self.pts_voxel_encoder, 'accepts_num_valid', False)
                    and getattr(self.pts_middle_encoder, 'accepts_num_valid', False))
        voxels, num_points, coors, _ = self.pts_voxel_layer.forward_batch(points, sync=sync)
        return voxels, num_points, coors

    @property
    def front_reads_counts(self):
        """True when the point-only front of a step reads voxel / site counts back to the host
        (the sparse-conv trunk: data-dependent level sizes); the PointPillars front does not."""
        return not (getattr(self.pts_voxel_encoder, 'accepts_num_valid', False)
                    and getattr(self.pts_middle_encoder, 'accepts_num_valid', False))

    @torch.no_grad()
    def prepare_inputs(self, points):
        """Voxelize ``points`` and build the sparse encoder's levels / rule books: everything of a
        step that depends on the points alone and nothing on the weights. ``forward_train`` accepts
        the result in place of ``points``; ``train.Runner`` calls this for the NEXT batch on a side
        stream, where its host reads of the counts wait only for these few kernels."""
        if isinstance(points, PreparedInputs):
            return points
        voxels, num_points, coors = self.voxelize(points)
        if hasattr(self.pts_middle_encoder, 'build_indices'):
            coors = self.pts_middle_encoder.build_indices(coors, len(points))
        return PreparedInputs(voxels, num_points, coors, len(points))

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None,
                      GGA_boxes_img=None, GGA_lidar2img=None, GGA_init_pseudo_labels=None, GGA_bdry_masks=None,
                      GGA_in_box_points=None, gt_labels=None, gt_bboxes=None, img=None, proposals=None,
                      gt_bboxes_ignore=None):
        img_feats, pts_feats = self.extract_feat(points, img=img, img_metas=img_metas)
        losses = dict()
        if pts_feats:
            losses.update(self.forward_pts_train(pts_feats, gt_bboxes_3d, gt_labels_3d, GGA_boxes_img,
                                                 GGA_lidar2img, GGA_init_pseudo_labels, GGA_bdry_masks,
                                                 GGA_in_box_points, img_metas, gt_bboxes_ignore))
        return losses

    def forward_pts_train(self, pts_feats, gt_bboxes_3d, gt_labels_3d, GGA_boxes_img, GGA_lidar2img,
                          GGA_init_pseudo_labels, GGA_bdry_masks, GGA_in_box_points, img_metas,
                          gt_bboxes_ignore=None):
        outs = self.pts_bbox_head(pts_feats)
        return self.pts_bbox_head.loss(gt_bboxes_3d, gt_labels_3d, outs, GGA_boxes_img, GGA_lidar2img,
                                       GGA_init_pseudo_labels, GGA_bdry_masks, GGA_in_box_points, img_metas)

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.forward_test(**kwargs)

    # ---- inference (mvx_two_stage_gga.py:407-423, centerpoint_gga.py:88-97, detectors/base.py:16-45)
    def forward_test(self, points, img_metas, img=None, **kwargs):
        for var, name in [(points, 'points'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        if len(points) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(points)}) != num of image meta ({len(img_metas)})')
        if len(points) == 1:
            return self.simple_test(points[0], img_metas[0], None if img is None else img[0], **kwargs)
        raise NotImplementedError('test-time augmentation is not on the GGA path')

    @torch.no_grad()
    def simple_test_pts(self, x, img_metas, rescale=False):
        from .box3d import bbox3d2result
        outs = self.pts_bbox_head(x)
        bbox_list = self.pts_bbox_head.get_bboxes(outs, img_metas, rescale=rescale)
        packed = getattr(bbox_list, 'packed', None)
        if packed is not None:
            # the batched post-processing left one set of batch tensors: three copies to the host for the whole batch
            # instead of three per frame (each a synchronisation); the per-frame results are host-side views
            boxes, scores, labels, counts = packed
            boxes, scores, labels = boxes.cpu(), scores.cpu(), labels.cpu()
            return [dict(boxes_3d=type(b)(boxes[i, :n], b.box_dim, with_yaw=b.with_yaw), scores_3d=scores[i, :n], labels_3d=labels[i, :n])
                    for i, (n, (b, _, _)) in enumerate(zip(counts, bbox_list))]
        return [bbox3d2result(bboxes, scores, labels) for bboxes, scores, labels in bbox_list]

    @torch.no_grad()
    def simple_test(self, points, img_metas, img=None, rescale=False):
        _, pts_feats = self.extract_feat(points, img=img, img_metas=img_metas)
        bbox_list = [dict() for _ in range(len(img_metas))]
        if pts_feats and self.with_pts_bbox:
            for result_dict, pts_bbox in zip(bbox_list, self.simple_test_pts(pts_feats, img_metas, rescale=rescale)):
                result_dict['pts_bbox'] = pts_bbox
        return bbox_list

    # ---- mmdet BaseDetector.train_step / _parse_losses (restated) -----------------
    def _parse_losses(self, losses):
        log_vars = OrderedDict()
        for name, value in losses.items():
            # (the mean of a 0-d tensor is the tensor: no reduce launch - and no division in backward - for the 18 scalar terms of a
            # CenterPoint-style head)
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean() if value.dim() else value
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() if v.dim() else v for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        pal = getattr(getattr(self, 'pts_bbox_head', None), 'pal_backprop', False)
        terms = [v for k, v in log_vars.items() if 'loss' in k or (pal and 'distance' in k)]
        loss = torch.stack(terms).sum()
        log_vars['loss'] = loss
        return loss, log_vars        # values stay on the device: no .item() sync per step

    def train_step(self, data, optimizer=None):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))


@DETECTORS.register_module()
class GGA(MVXTwoStageDetector_GGA):
    """The GGA detector (centerpoint_gga.py:10-86): CenterPoint-style single stage on top of
    ``MVXTwoStageDetector_GGA``."""

    @property
    def with_velocity(self):
        return self.pts_bbox_head is not None and self.pts_bbox_head.with_velocity
