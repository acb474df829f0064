"""``LiDARInstance3DBoxes`` — the slice of the reference's box structure the GGA head and its
post-processing touch (mmdet3d/core/bbox/structures/base_box3d.py, lidar_box3d.py): tensor
``(x, y, z_bottom, dx, dy, dz, yaw)``, ``bev``, centres, ``overlaps``, ``points_in_boxes_*``."""
import math

import numpy as np

import torch

from . import functional as F
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
            shift = F.const_tensor([0.5 - origin[0], 0.5 - origin[1], 0.0 - origin[2]], self.tensor.device, self.tensor.dtype)
            self.tensor[:, :3] += self.tensor[:, 3:6] * shift

    @classmethod
    def wrap(cls, tensor, box_dim=7, with_yaw=True):
        """A box object OVER ``tensor`` [N, box_dim] (no copy: the constructor clones, one launch per frame in a loop over a
        batch's detections); origin (0.5, 0.5, 0)."""
        obj = cls.__new__(cls)
        obj.tensor, obj.box_dim, obj.with_yaw = tensor, box_dim, with_yaw
        return obj

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


# ----------------------------------------------------------------------------- camera boxes (mono3d heads)
def rotation_about_y(points, angles):
    """``rotation_3d_in_axis(points [N,M,3], angles [N], axis=1)`` (core/bbox/structures/utils.py:28-117)."""
    c, s = torch.cos(angles), torch.sin(angles)
    o, z = torch.ones_like(c), torch.zeros_like(c)
    rot_t = torch.stack([torch.stack([c, z, -s]), torch.stack([z, o, z]), torch.stack([s, z, c])])
    return torch.einsum('aij,jka->aik', points, rot_t)


def _matrix(m, like):
    """A small host matrix (camera intrinsics of a sample) as a device tensor of ``like``'s type, cached by value."""
    if isinstance(m, torch.Tensor):
        return m.to(device=like.device, dtype=like.dtype)
    return F.const_tensor(np.ascontiguousarray(m), like.device, like.dtype)


def points_cam2img(points_3d, proj_mat, with_depth=False):
    """core/bbox/structures/utils.py:173-214."""
    points_shape = list(points_3d.shape)
    points_shape[-1] = 1
    proj_mat = _matrix(proj_mat, points_3d)
    d1, d2 = proj_mat.shape[:2]
    if d1 == 3:
        ex = torch.eye(4, device=proj_mat.device, dtype=proj_mat.dtype)
        ex[:d1, :d2] = proj_mat
        proj_mat = ex
    p4 = torch.cat([points_3d, points_3d.new_ones(points_shape)], dim=-1)
    p2 = p4 @ proj_mat.T
    res = p2[..., :2] / p2[..., 2:3]
    return torch.cat([res, p2[..., 2:3]], dim=-1) if with_depth else res


def points_img2cam(points, cam2img):
    """core/bbox/structures/utils.py:217-248: (u, v, depth) -> camera xyz."""
    xys, depths = points[:, :2], points[:, 2].view(-1, 1)
    un = torch.cat([xys * depths, depths], dim=1)
    if isinstance(cam2img, torch.Tensor):
        cam2img = cam2img.to(device=points.device, dtype=points.dtype)
        pad = torch.eye(4, dtype=xys.dtype, device=xys.device)
        pad[:cam2img.shape[0], :cam2img.shape[1]] = cam2img
        inv = torch.inverse(pad).transpose(0, 1)
    else:
        # a host matrix (the sample's intrinsics): inverted on the host and cached - torch.inverse on the device waits
        # for the device on every call (the library's LU reports its status to the host), 24 times per PGD step
        m = np.asarray(cam2img, dtype=np.float64)
        pad = np.eye(4)
        pad[:m.shape[0], :m.shape[1]] = m
        inv = F.const_tensor(np.ascontiguousarray(np.linalg.inv(pad).T.astype(np.float32)), points.device, points.dtype)
    homo = torch.cat([un, xys.new_ones((un.shape[0], 1))], dim=1)
    return torch.mm(homo, inv)[:, :3]


class CameraInstance3DBoxes:
    """The slice of core/bbox/structures/cam_box3d.py the mono3d heads use: tensor
    (x, y, z, x_size, y_size, z_size, yaw) with the bottom centre as origin (0.5, 1.0, 0.5), ``corners``."""
    YAW_AXIS = 1

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 1.0, 0.5)):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device('cpu')
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim))
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim, tensor.size()
        self.box_dim, self.with_yaw = box_dim, with_yaw
        self.tensor = tensor.clone()
        if origin != (0.5, 1.0, 0.5):
            shift = F.const_tensor([0.5 - origin[0], 1.0 - origin[1], 0.5 - origin[2]], self.tensor.device, self.tensor.dtype)
            self.tensor[:, :3] += self.tensor[:, 3:6] * shift

    dims = property(lambda self: self.tensor[:, 3:6])
    yaw = property(lambda self: self.tensor[:, 6])

    @property
    def bev(self):
        """XYWHR in the ground plane (x, z, x_size, z_size, -yaw): the camera's gravity axis points down
        (cam_box3d.py:159-168)."""
        bev = self.tensor[:, [0, 2, 3, 5, 6]].clone()
        bev[:, -1] = -bev[:, -1]
        return bev

    def __len__(self):
        return self.tensor.shape[0]

    def to(self, device):
        return CameraInstance3DBoxes(self.tensor.to(device), box_dim=self.box_dim, with_yaw=self.with_yaw)

    @property
    def corners(self):
        if self.tensor.numel() == 0:
            return torch.empty([0, 8, 3], device=self.tensor.device)
        dims = self.dims
        idx = [[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [1, 1, 0]]     # unravel_index order [0,1,3,2,4,5,7,6]
        corners_norm = F.const_tensor([[a - 0.5, b - 1.0, c - 0.5] for a, b, c in idx], dims.device, dims.dtype)
        corners = dims.view([-1, 1, 3]) * corners_norm.reshape([1, 8, 3])
        corners = rotation_about_y(corners, self.tensor[:, 6])
        corners += self.tensor[:, :3].view(-1, 1, 3)
        return corners
