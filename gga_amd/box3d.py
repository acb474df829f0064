"""``LiDARInstance3DBoxes`` — the slice of the reference's box structure the GGA head and its
post-processing touch (mmdet3d/core/bbox/structures/base_box3d.py, lidar_box3d.py): tensor
``(x, y, z_bottom, dx, dy, dz, yaw)``, ``bev``, centres, ``overlaps``, ``points_in_boxes_*``."""
import math

import torch

from . import ops


class LiDARInstance3DBoxes:
    YAW_AXIS = 2

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        if isinstance(tensor, torch.Tensor):
            device = tensor.device
        else:
            device = torch.device('cpu')
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim, tensor.size()
        if tensor.shape[-1] == 6:
            assert box_dim == 6
            tensor = torch.cat((tensor, tensor.new_zeros(tensor.shape[0], 1)), dim=-1)
            self.box_dim, self.with_yaw = box_dim + 1, False
        else:
            self.box_dim, self.with_yaw = box_dim, with_yaw
        self.tensor = tensor.clone()
        if origin != (0.5, 0.5, 0):
            dst = self.tensor.new_tensor((0.5, 0.5, 0))
            src = self.tensor.new_tensor(origin)
            self.tensor[:, :3] += self.tensor[:, 3:6] * (dst - src)

    volume = property(lambda self: self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5])
    dims = property(lambda self: self.tensor[:, 3:6])
    yaw = property(lambda self: self.tensor[:, 6])
    height = property(lambda self: self.tensor[:, 5])
    top_height = property(lambda self: self.bottom_height + self.height)
    bottom_height = property(lambda self: self.tensor[:, 2])
    center = property(lambda self: self.bottom_center)
    bottom_center = property(lambda self: self.tensor[:, :3])
    bev = property(lambda self: self.tensor[:, [0, 1, 3, 4, 6]])
    device = property(lambda self: self.tensor.device)

    @property
    def gravity_center(self):
        bc = self.bottom_center
        gc = torch.zeros_like(bc)
        gc[:, :2] = bc[:, :2]
        gc[:, 2] = bc[:, 2] + self.tensor[:, 5] * 0.5
        return gc

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        t = self.tensor[item]
        if t.dim() == 1:
            t = t.view(1, -1)
        return type(self)(t, box_dim=self.box_dim, with_yaw=self.with_yaw)

    def __repr__(self):
        return self.__class__.__name__ + '(\n    ' + str(self.tensor) + ')'

    def to(self, device):
        return type(self)(self.tensor.to(device), box_dim=self.box_dim, with_yaw=self.with_yaw)

    def clone(self):
        return type(self)(self.tensor.clone(), box_dim=self.box_dim, with_yaw=self.with_yaw)

    def new_box(self, data):
        """base_box3d.py:488-508: a box object of the same type / device from array-like ``data``."""
        t = self.tensor.new_tensor(data) if not isinstance(data, torch.Tensor) else data.to(self.device)
        return type(self)(t, box_dim=self.box_dim, with_yaw=self.with_yaw)

    def limit_yaw(self, offset=0.5, period=math.pi):
        """base_box3d.py:272-279 / structures/utils.py:11-25: yaw -> [-offset*period, (1-offset)*period)."""
        yaw = self.tensor[:, 6]
        self.tensor[:, 6] = yaw - torch.floor(yaw / period + offset) * period

    @classmethod
    def height_overlaps(cls, boxes1, boxes2, mode='iou'):
        top = torch.min(boxes1.top_height.view(-1, 1), boxes2.top_height.view(1, -1))
        bot = torch.max(boxes1.bottom_height.view(-1, 1), boxes2.bottom_height.view(1, -1))
        return torch.clamp(top - bot, min=0)

    @classmethod
    def overlaps(cls, boxes1, boxes2, mode='iou'):
        """3D IoU / IoF of two box sets (base_box3d.py:440-500) on the HIP rotated-IoU kernel."""
        assert type(boxes1) == type(boxes2), f'"boxes1" and "boxes2" should be in the same type, got {type(boxes1)} and {type(boxes2)}.'
        assert mode in ['iou', 'iof']
        rows, cols = len(boxes1), len(boxes2)
        if rows * cols == 0:
            return boxes1.tensor.new(rows, cols)
        overlaps_h = cls.height_overlaps(boxes1, boxes2)
        iou2d = ops.box_iou_rotated(boxes1.bev, boxes2.bev)
        areas1 = (boxes1.bev[:, 2] * boxes1.bev[:, 3]).unsqueeze(1).expand(rows, cols)
        areas2 = (boxes2.bev[:, 2] * boxes2.bev[:, 3]).unsqueeze(0).expand(rows, cols)
        overlaps_bev = iou2d * (areas1 + areas2) / (1 + iou2d)
        overlaps_3d = overlaps_bev.to(boxes1.device) * overlaps_h
        volume1, volume2 = boxes1.volume.view(-1, 1), boxes2.volume.view(1, -1)
        if mode == 'iou':
            return overlaps_3d / torch.clamp(volume1 + volume2 - overlaps_3d, min=1e-8)
        return overlaps_3d / torch.clamp(volume1, min=1e-8)

    def points_in_boxes_part(self, points, boxes_override=None):
        boxes = boxes_override if boxes_override is not None else self.tensor
        if points.dim() == 2:
            points = points.unsqueeze(0)
        return ops.points_in_boxes_part(points[..., :3], boxes[:, :7].unsqueeze(0).to(points.device)).squeeze(0)

    def points_in_boxes_all(self, points, boxes_override=None):
        boxes = boxes_override if boxes_override is not None else self.tensor
        pts = points.clone()[..., :3]
        if pts.dim() == 2:
            pts = pts.unsqueeze(0)
        else:
            assert pts.dim() == 3 and pts.shape[0] == 1
        return ops.points_in_boxes_all(pts, boxes[:, :7].to(pts.device).unsqueeze(0)).squeeze(0)


def bbox3d2result(bboxes, scores, labels, attrs=None):
    """mmdet3d/core/bbox/transforms.py:bbox3d2result — detections to the CPU result dict."""
    result = dict(boxes_3d=bboxes.to('cpu'), scores_3d=scores.cpu(), labels_3d=labels.cpu())
    if attrs is not None:
        result['attrs_3d'] = attrs.cpu()
    return result
