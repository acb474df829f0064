"""``CenterPointBBoxCoder`` — registry name and constructor of the reference's coder
(mmdet3d/core/bbox/coders/centerpoint_bbox_coders.py:8-229). The train step only
constructs it (centerpoint_head_gga.py:86); ``decode`` serves the inference /
pseudo-label path.

``decode`` here is a single flat top-k over (class, cell) — the same set the reference
obtains with its per-class top-k followed by a cross-class top-k (:63-96) — then one
gather of the regression channels at the winning cells.
"""
import torch

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
        lim = torch.tensor(self.post_center_range, device=heat.device)
        keep = (boxes[..., :3] >= lim[:3]).all(2) & (boxes[..., :3] <= lim[3:]).all(2)
        if self.score_threshold is not None:
            keep &= score > self.score_threshold
        return [dict(bboxes=boxes[i, keep[i]], scores=score[i, keep[i]], labels=label[i, keep[i]].float())
                for i in range(B)]
