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
        self.pc_range, self.voxel_size, self.out_size_factor = pc_range, voxel_size, out_size_factor
        self.post_center_range, self.max_num, self.score_threshold, self.code_size = post_center_range, max_num, score_threshold, code_size

    def encode(self):
        pass

    @staticmethod
    def _at(fmap, cells):
        """fmap [B,C,H,W], cells [B,K] -> [B,K,C]"""
        B, C = fmap.shape[:2]
        return fmap.reshape(B, C, -1).gather(2, cells.unsqueeze(1).expand(B, C, cells.shape[1])).transpose(1, 2)

    def decode_dense(self, heat, rot_sine, rot_cosine, hei, dim, vel, reg=None, task_id=-1):
        """The top-``max_num`` cells of every frame, decoded, BEFORE the masks: boxes [B, K, code], scores [B, K] (descending),
        labels [B, K] (class within the task, float). ``decode`` masks them per frame; the head's batched post-processing
        (``functional.centerpoint_detect``) applies the same masks on the device for all frames and tasks at once."""
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
        return torch.cat(parts, dim=2), score, label.float()

    def decode(self, heat, rot_sine, rot_cosine, hei, dim, vel, reg=None, task_id=-1):
        boxes, score, label = self.decode_dense(heat, rot_sine, rot_cosine, hei, dim, vel, reg=reg, task_id=task_id)
        lim = F.const_tensor(list(self.post_center_range), heat.device)
        keep = (boxes[..., :3] >= lim[:3]).all(2) & (boxes[..., :3] <= lim[3:]).all(2)
        if self.score_threshold is not None:
            keep &= score > self.score_threshold
        return [dict(bboxes=boxes[i, keep[i]], scores=score[i, keep[i]], labels=label[i, keep[i]])
                for i in range(heat.shape[0])]


# ----------------------------------------------------------------------------- mono3d coders (PGD / FCOS3D)
import numpy as np  # noqa: E402
from torch.nn import functional as TF  # noqa: E402


def limit_period(val, offset=0.5, period=np.pi):
    return val - torch.floor(val / period + offset) * period


@BBOX_CODERS.register_module()
class FCOS3DBBoxCoder:
    """mmdet3d/core/bbox/coders/fcos3d_bbox_coder.py:10-127 (interface: ``decode`` / ``decode_yaw`` and the constructor keys).
    Raw regression maps [B, C, H, W] with channels (dx, dy | depth | w, h, l | yaw ...) -> metric quantities: a learnable
    factor per group and level, ``exp`` (or an affine prior) for depth, ``exp`` (times a class prior) for the sizes; offsets
    are in units of the level's stride while training and in pixels at test time. Written without in-place updates: the
    result is a new tensor assembled from the channel groups."""

    def __init__(self, base_depths=None, base_dims=None, code_size=7, norm_on_bbox=True):
        self.base_depths, self.base_dims, self.bbox_code_size, self.norm_on_bbox = base_depths, base_dims, code_size, norm_on_bbox

    def encode(self, gt_bboxes_3d, gt_labels_3d, gt_bboxes, gt_labels):
        """Targets are built by the head (``gga_fcos3d_targets``); nothing to encode."""

    @staticmethod
    def _class_prior(table, cls_score):
        """Per-location rows of a per-class table, chosen by the best-scoring class: [B, H, W, ...] -> channels first."""
        assert len(table) == cls_score.shape[1], 'one prior per class'
        best = cls_score.argmax(dim=1)
        return F.const_tensor(table, cls_score.device, cls_score.dtype)[best].movedim(-1, 1)

    def decode(self, bbox, scale, stride, training, cls_score=None):
        assert self.norm_on_bbox is True
        offset_factor, depth_factor, size_factor = scale[0], scale[1], scale[2]
        offset = offset_factor(bbox[:, 0:2]).float()
        depth = depth_factor(bbox[:, 2:3]).float()
        size = size_factor(bbox[:, 3:6]).float().exp()
        if self.base_depths is None:
            depth = depth.exp()
        elif len(self.base_depths) == 1:
            (mean, std), = self.base_depths
            depth = mean + depth * std
        else:
            prior = self._class_prior(self.base_depths, cls_score)                # [B, 2, H, W]: mean, std of the class
            depth = prior[:, 0:1] + depth * prior[:, 1:2]
        if self.base_dims is not None:
            size = self._class_prior(self.base_dims, cls_score) * size
        if not training:
            offset = offset * stride
        return torch.cat([offset.to(bbox.dtype), depth.to(bbox.dtype), size.to(bbox.dtype), bbox[:, 6:]], dim=1)

    @staticmethod
    def decode_yaw(bbox, centers2d, dir_cls, dir_offset, cam2img):
        """Local yaw in [offset, offset + pi) plus the direction bin's half turn, then + the viewing angle of the projected
        centre (local -> global yaw). ``bbox`` [n, >= 7] is updated in its yaw column and returned."""
        yaw = bbox[:, 6]
        if len(bbox):
            yaw = limit_period(yaw - dir_offset, 0, np.pi) + dir_offset + np.pi * dir_cls.to(bbox.dtype)
        view = torch.atan2(centers2d[:, 0] - cam2img[0, 2], cam2img[0, 0])
        bbox[:, 6] = view + yaw
        return bbox


@BBOX_CODERS.register_module()
class PGDBBoxCoder(FCOS3DBBoxCoder):
    """mmdet3d/core/bbox/coders/pgd_bbox_coder.py:10-128: key-point / 2D-distance channels and the
    probabilistic depth read-out."""

    def decode_2d(self, bbox, scale, stride, max_regress_range, training, pred_keypoints=False, pred_bbox2d=True):
        """The 16 key-point offsets (tanh of the scaled raw values; times the level's regress range at test time) right after
        the 3D code, and the four distances to the 2D box sides in the last four channels (ReLU when normalised by the
        stride - times the stride at test time -, ``exp`` otherwise)."""
        cs = self.bbox_code_size
        head, keys, mid, sides = bbox[:, :cs], bbox[:, cs:cs + 16] if pred_keypoints else None, None, None
        if pred_bbox2d:
            sides = scale[-1](bbox[:, -4:]).float()
            sides = TF.relu(sides) if self.norm_on_bbox else sides.exp()
            if self.norm_on_bbox and not training:
                sides = sides * stride
        if pred_keypoints:
            keys = torch.tanh(scale[3](keys).float())
            if self.norm_on_bbox and not training:
                keys = keys * max_regress_range
        lo = cs + 16 if pred_keypoints else cs
        hi = bbox.shape[1] - 4 if pred_bbox2d else bbox.shape[1]
        mid = bbox[:, lo:hi]
        parts = [head] + ([keys.to(bbox.dtype)] if pred_keypoints else []) + [mid] + ([sides.to(bbox.dtype)] if pred_bbox2d else [])
        return torch.cat(parts, dim=1)

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
