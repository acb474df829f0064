"""``CenterPointBBoxCoder`` — registry name and constructor of the reference's coder
(mmdet3d/core/bbox/coders/centerpoint_bbox_coders.py:8-229). The train step only
constructs it (centerpoint_head_gga.py:86); ``decode`` serves the inference /
pseudo-label path.

``decode`` here is a single flat top-k over (class, cell) — the same set the reference
obtains with its per-class top-k followed by a cross-class top-k (:63-96) — then one
gather of the regression channels at the winning cells.
"""
import torch

from . import functional as F
from .registry import BBOX_CODERS


@BBOX_CODERS.register_module()
class CenterPointBBoxCoder:
    def __init__(self, pc_range, out_size_factor, voxel_size, post_center_range=None, max_num=100,
                 score_threshold=None, code_size=9):
        self.pc_range = pc_range
        self.out_size_factor = out_size_factor
        self.voxel_size = voxel_size
        self.post_center_range = post_center_range
        self.max_num = max_num
        self.score_threshold = score_threshold
        self.code_size = code_size

    def encode(self):
        pass

    @staticmethod
    def _at(fmap, cells):
        """fmap [B,C,H,W], cells [B,K] -> [B,K,C]"""
        B, C = fmap.shape[:2]
        return fmap.reshape(B, C, -1).gather(2, cells.unsqueeze(1).expand(B, C, cells.shape[1])).transpose(1, 2)

    def decode(self, heat, rot_sine, rot_cosine, hei, dim, vel, reg=None, task_id=-1):
        if self.post_center_range is None:
            raise NotImplementedError('Need to reorganize output as a batch, only support '
                                      'post_center_range is not None for now!')
        B, ncls, H, W = heat.shape
        K = self.max_num
        score, flat = heat.reshape(B, -1).topk(K)
        label = torch.div(flat, H * W, rounding_mode='floor')
        cell = flat % (H * W)
        ys = torch.div(cell, W, rounding_mode='floor').float().unsqueeze(2)
        xs = (cell % W).float().unsqueeze(2)
        if reg is not None:
            off = self._at(reg, cell)
            xs, ys = xs + off[..., 0:1], ys + off[..., 1:2]
        else:
            xs, ys = xs + 0.5, ys + 0.5
        rot = torch.atan2(self._at(rot_sine, cell), self._at(rot_cosine, cell))
        xs = xs * self.out_size_factor * self.voxel_size[0] + self.pc_range[0]
        ys = ys * self.out_size_factor * self.voxel_size[1] + self.pc_range[1]
        parts = [xs, ys, self._at(hei, cell), self._at(dim, cell), rot]
        if vel is not None:
            parts.append(self._at(vel, cell))
        boxes = torch.cat(parts, dim=2)
        lim = F.const_tensor(list(self.post_center_range), heat.device)
        keep = (boxes[..., :3] >= lim[:3]).all(2) & (boxes[..., :3] <= lim[3:]).all(2)
        if self.score_threshold is not None:
            keep &= score > self.score_threshold
        return [dict(bboxes=boxes[i, keep[i]], scores=score[i, keep[i]], labels=label[i, keep[i]].float())
                for i in range(B)]


# ----------------------------------------------------------------------------- mono3d coders (PGD / FCOS3D)
import numpy as np  # noqa: E402
from torch.nn import functional as TF  # noqa: E402


def limit_period(val, offset=0.5, period=np.pi):
    return val - torch.floor(val / period + offset) * period


@BBOX_CODERS.register_module()
class FCOS3DBBoxCoder:
    """mmdet3d/core/bbox/coders/fcos3d_bbox_coder.py:10-127: per-level learnable scales, depth / size priors,
    stride un-normalisation at test time, local yaw -> global yaw."""

    def __init__(self, base_depths=None, base_dims=None, code_size=7, norm_on_bbox=True):
        self.base_depths, self.base_dims, self.bbox_code_size, self.norm_on_bbox = base_depths, base_dims, code_size, norm_on_bbox

    def encode(self, gt_bboxes_3d, gt_labels_3d, gt_bboxes, gt_labels):
        pass

    def decode(self, bbox, scale, stride, training, cls_score=None):
        scale_offset, scale_depth, scale_size = scale[0:3]
        clone_bbox = bbox.clone()
        bbox[:, :2] = scale_offset(clone_bbox[:, :2]).float()
        bbox[:, 2] = scale_depth(clone_bbox[:, 2]).float()
        bbox[:, 3:6] = scale_size(clone_bbox[:, 3:6]).float()
        if self.base_depths is None:
            bbox[:, 2] = bbox[:, 2].exp()
        elif len(self.base_depths) == 1:
            mean, std = self.base_depths[0]
            bbox[:, 2] = mean + bbox.clone()[:, 2] * std
        else:
            assert len(self.base_depths) == cls_score.shape[1]
            indices = cls_score.max(dim=1)[1]
            priors = F.const_tensor(self.base_depths, cls_score.device, cls_score.dtype)[indices, :].permute(0, 3, 1, 2)
            bbox[:, 2] = priors[:, 0] + bbox.clone()[:, 2] * priors[:, 1]
        bbox[:, 3:6] = bbox[:, 3:6].exp()
        if self.base_dims is not None:
            assert len(self.base_dims) == cls_score.shape[1]
            indices = cls_score.max(dim=1)[1]
            size_priors = F.const_tensor(self.base_dims, cls_score.device, cls_score.dtype)[indices, :].permute(0, 3, 1, 2)
            bbox[:, 3:6] = size_priors * bbox.clone()[:, 3:6]
        assert self.norm_on_bbox is True
        if not training:
            bbox[:, :2] *= stride
        return bbox

    @staticmethod
    def decode_yaw(bbox, centers2d, dir_cls, dir_offset, cam2img):
        if bbox.shape[0] > 0:
            dir_rot = limit_period(bbox[..., 6] - dir_offset, 0, np.pi)
            bbox[..., 6] = dir_rot + dir_offset + np.pi * dir_cls.to(bbox.dtype)
        bbox[:, 6] = torch.atan2(centers2d[:, 0] - cam2img[0, 2], cam2img[0, 0]) + bbox[:, 6]
        return bbox


@BBOX_CODERS.register_module()
class PGDBBoxCoder(FCOS3DBBoxCoder):
    """mmdet3d/core/bbox/coders/pgd_bbox_coder.py:10-128: key-point / 2D-distance channels and the
    probabilistic depth read-out."""

    def decode_2d(self, bbox, scale, stride, max_regress_range, training, pred_keypoints=False, pred_bbox2d=True):
        clone_bbox = bbox.clone()
        cs = self.bbox_code_size
        if pred_keypoints:
            bbox[:, cs:cs + 16] = torch.tanh(scale[3](clone_bbox[:, cs:cs + 16]).float())
        if pred_bbox2d:
            bbox[:, -4:] = scale[-1](clone_bbox[:, -4:]).float()
        if self.norm_on_bbox:
            if pred_bbox2d:
                bbox[:, -4:] = TF.relu(bbox.clone()[:, -4:])
            if not training:
                if pred_keypoints:
                    bbox[:, cs:cs + 16] *= max_regress_range
                if pred_bbox2d:
                    bbox[:, -4:] *= stride
        elif pred_bbox2d:
            bbox[:, -4:] = bbox.clone()[:, -4:].exp()
        return bbox

    def decode_prob_depth(self, depth_cls_preds, depth_range, depth_unit, division, num_depth_cls):
        split = F.const_tensor(list(range(num_depth_cls)), depth_cls_preds.device, depth_cls_preds.dtype).reshape([1, -1])
        prob = TF.softmax(depth_cls_preds.clone(), dim=-1)
        if division == 'uniform':
            return (prob * (depth_unit * split)).sum(dim=-1)
        if division == 'linear':
            mult = depth_range[0] + (depth_range[1] - depth_range[0]) / (num_depth_cls * (num_depth_cls - 1)) * (split * (split + 1))
            return (prob * mult).sum(dim=-1)
        start, end = max(depth_range[0], 1), depth_range[1]
        log_mult = np.log(start) + split * np.log(end / start) / (num_depth_cls - 1)
        if division == 'log':
            return (prob * log_mult.exp()).sum(dim=-1)
        if division == 'loguniform':
            return (prob * log_mult).sum(dim=-1).exp()
        raise NotImplementedError
