"""``Voxelization`` — same constructor / forward contract as ``mmcv.ops.Voxelization``
as the reference uses it (mmdet3d/models/detectors/mvx_two_stage_gga.py:43,225:
``Voxelization(**pts_voxel_layer)``, ``voxels, coors, num_points = layer(points)``),
plus ``forward_batch`` which voxelizes every frame of the batch in ONE call of the HIP
path instead of the reference's per-frame Python loop."""
import torch
from torch import nn

from . import functional as F


class Voxelization(nn.Module):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000,
                 deterministic=True):
        super().__init__()
        self.voxel_size = list(voxel_size)
        self.point_cloud_range = list(point_cloud_range)
        self.max_num_points = int(max_num_points)
        self.max_voxels = tuple(max_voxels) if isinstance(max_voxels, (tuple, list)) else (max_voxels, max_voxels)
        self.deterministic = deterministic      # the HIP path is always deterministic
        prm = F.voxel_params(self.voxel_size, self.point_cloud_range, 1, 1)
        grid = F.voxel_grid_size(prm)
        self.grid_size = torch.tensor(grid)
        self.pcd_shape = [*grid[:2], 1][::-1]
        if self.max_num_points == -1 or self.max_voxels[0] == -1:
            raise NotImplementedError('dynamic voxelization (max_num_points=-1) is not on the GGA path')

    def _cap(self):
        return self.max_voxels[0] if self.training else self.max_voxels[1]

    def forward(self, input):
        """points [N, C] -> voxels [M, P, C], coors [M, 3] (z, y, x), num_points [M]."""
        voxels, num_points, coors, _ = F.hard_voxelize_batch([input], self.voxel_size, self.point_cloud_range,
                                                             self.max_num_points, self._cap())
        return voxels, coors[:, 1:].contiguous(), num_points

    def forward_batch(self, points, sync=True):
        """list of [N_b, C] -> voxels [SM, P, C], num_points [SM], coors [SM, 4] (b, z, y, x),
        voxel_num [B+1] — what ``MVXTwoStageDetector_GGA.voxelize`` assembles."""
        return F.hard_voxelize_batch(points, self.voxel_size, self.point_cloud_range, self.max_num_points,
                                     self._cap(), sync=sync)

    def forward_prepared(self, prep, sync=True):
        """``forward_batch`` for the output of ``functional.points_prepare_batch`` (device-resident
        frames with device-side point counts)."""
        return F.hard_voxelize_prepared(prep, self.voxel_size, self.point_cloud_range, self.max_num_points,
                                        self._cap(), sync=sync)

    def __repr__(self):
        return (f'{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range='
                f'{self.point_cloud_range}, max_num_points={self.max_num_points}, max_voxels={self.max_voxels}, '
                f'deterministic={self.deterministic})')
