"""Middle encoders: ``PointPillarsScatter`` (HIP scatter kernel; reference
mmdet3d/models/middle_encoders/pillar_scatter.py:9-102). ``SparseEncoder`` lives in
``sparse_encoder.py``."""
from torch import nn

from . import functional as F
from .registry import MIDDLE_ENCODERS


@MIDDLE_ENCODERS.register_module()
class PointPillarsScatter(nn.Module):
    """``forward(voxel_features [M,C], coors [M,4] (b,z,y,x), batch_size) -> [B,C,ny,nx]``.

    ``channels_last=True`` (not in the reference) returns the same logical NCHW tensor in
    NHWC memory, so every pillar is one contiguous row for the scatter and MIOpen gets an
    NHWC input; values are identical, only strides differ.
    """

    def __init__(self, in_channels, output_shape, channels_last=False):
        super().__init__()
        self.output_shape = output_shape
        self.ny = output_shape[0]
        self.nx = output_shape[1]
        self.in_channels = in_channels
        self.channels_last = channels_last
        self.unique_coors = True
        self.fp16_enabled = False

    accepts_num_valid = True

    def forward(self, voxel_features, coors, batch_size=None, num_valid=None):
        if num_valid is None:
            num_valid = F.num_valid_of(coors)       # capacity-sized buffers of the sync-free voxelizer
            if num_valid is not None and voxel_features.shape[0] != coors.shape[0]:
                m = voxel_features.shape[0]         # the encoder already trimmed to the exact rows
                coors, num_valid = coors[:m], None
        if batch_size is None:
            # forward_single of the reference: one sample, batch index ignored
            coors = coors.clone()
            coors[:, 0] = 0
            batch_size = 1
        # coordinates come from hard voxelization: one pillar per cell (the reference's index_put
        # is undefined for repeated cells on the GPU); set unique_coors = False for arbitrary input
        return F.pillar_scatter(voxel_features, coors, int(batch_size), self.ny, self.nx,
                                channels_last=self.channels_last, num_valid=num_valid, unique=self.unique_coors)
