"""Voxel encoders of the GGA configs: ``HardSimpleVFE`` (SECOND trunk,
mmdet3d/models/voxel_encoders/voxel_encoder.py:13-45) and ``PillarFeatureNet`` +
``PFNLayer`` (PointPillars trunk, voxel_encoders/pillar_encoder.py:12-159,
voxel_encoders/utils.py:107-182). Same constructor arguments, parameter names and
``forward(features, num_points, coors)`` signature."""
import torch
from torch import nn
from torch.nn import functional as TF

from . import functional as F
from .cnn import build_norm_layer
from .registry import VOXEL_ENCODERS


@VOXEL_ENCODERS.register_module()
class HardSimpleVFE(nn.Module):
    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features
        self.fp16_enabled = False

    def forward(self, features, num_points, coors):
        return F.voxel_mean(features, num_points, self.num_features)


def get_paddings_indicator(actual_num, max_num, axis=0):
    actual_num = torch.unsqueeze(actual_num, axis + 1)
    shape = [1] * len(actual_num.shape)
    shape[axis + 1] = -1
    max_num = torch.arange(max_num, dtype=torch.int, device=actual_num.device).view(shape)
    return actual_num.int() > max_num


class PFNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01),
                 last_layer=False, mode='max'):
        super().__init__()
        self.fp16_enabled = False
        self.name = 'PFNLayer'
        self.last_vfe = last_layer
        if not self.last_vfe:
            out_channels = out_channels // 2
        self.units = out_channels
        self.norm = build_norm_layer(norm_cfg, self.units)[1]
        self.linear = nn.Linear(in_channels, self.units, bias=False)
        assert mode in ['max', 'avg']
        self.mode = mode

    def forward(self, inputs, num_voxels=None, aligned_distance=None):
        x = self.linear(inputs)
        x = self.norm(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()
        x = TF.relu(x)
        if aligned_distance is not None:
            x = x.mul(aligned_distance.unsqueeze(-1))
        if self.mode == 'max':
            x_max = torch.max(x, dim=1, keepdim=True)[0]
        else:
            x_max = x.sum(dim=1, keepdim=True) / num_voxels.type_as(inputs).view(-1, 1, 1)
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max.repeat(1, inputs.shape[1], 1)], dim=2)


@VOXEL_ENCODERS.register_module()
class PillarFeatureNet(nn.Module):
    def __init__(self, in_channels=4, feat_channels=(64, ), with_distance=False, with_cluster_center=True,
                 with_voxel_center=True, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), mode='max', legacy=True):
        super().__init__()
        assert len(feat_channels) > 0
        self.legacy = legacy
        if with_cluster_center:
            in_channels += 3
        if with_voxel_center:
            in_channels += 3
        if with_distance:
            in_channels += 1
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.fp16_enabled = False
        self.in_channels = in_channels
        feat_channels = [in_channels] + list(feat_channels)
        self.pfn_layers = nn.ModuleList([
            PFNLayer(feat_channels[i], feat_channels[i + 1], norm_cfg=norm_cfg,
                     last_layer=(i >= len(feat_channels) - 2), mode=mode)
            for i in range(len(feat_channels) - 1)])
        self.vx, self.vy, self.vz = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.point_cloud_range = point_cloud_range

    def _fusable(self, features):
        l0 = self.pfn_layers[0]
        return (features.is_cuda and len(self.pfn_layers) == 1 and l0.units == 64 and l0.mode == 'max'
                and self.legacy and self._with_cluster_center and self._with_voxel_center
                and not self._with_distance and features.shape[-1] == 4 and features.shape[1] <= 254
                and features.shape[0] > 0 and isinstance(l0.norm, nn.BatchNorm1d) and l0.norm.affine
                and l0.norm.track_running_stats and l0.norm.momentum is not None and features.dtype == torch.float32)

    accepts_num_valid = True        # capacity-sized inputs with a device-side count (detectors.voxelize)

    def forward(self, features, num_points, coors):
        if self._fusable(features):
            # one fused HIP pass (gga_amd/csrc/pfn.hip) instead of the [M,P,64] eager pipeline
            l0 = self.pfn_layers[0]
            bn = l0.norm
            prm = F.pfn_params((self.vx, self.vy, self.vz), (self.x_offset, self.y_offset, self.z_offset),
                               bn.eps, bn.momentum, self.training)
            if self.training:
                F.count_batch(bn)
            return F.fused_pfn(features, num_points.int(), coors.int(), l0.linear.weight, bn.weight, bn.bias,
                               bn.running_mean, bn.running_var, prm, num_valid=F.num_valid_of(coors))
        if F.num_valid_of(coors) is not None:     # capacity-sized buffers: the eager ops need the exact rows
            m = int(F.num_valid_of(coors).item())
            features, num_points, coors = features[:m], num_points[:m], coors[:m]
        return self.forward_eager(features, num_points, coors)

    def forward_eager(self, features, num_points, coors):
        """The reference's op sequence in eager PyTorch (general configurations)."""
        features_ls = [features]
        if self._with_cluster_center:
            points_mean = features[:, :, :3].sum(dim=1, keepdim=True) / num_points.type_as(features).view(-1, 1, 1)
            features_ls.append(features[:, :, :3] - points_mean)
        if self._with_voxel_center:
            centre = torch.stack([coors[:, 3].type_as(features) * self.vx + self.x_offset,
                                  coors[:, 2].type_as(features) * self.vy + self.y_offset,
                                  coors[:, 1].type_as(features) * self.vz + self.z_offset], 1).unsqueeze(1)
            f_center = features[:, :, :3] - centre
            if self.legacy:
                # legacy=True subtracts in place through a view (pillar_encoder.py:129-139): the
                # first three input channels ARE the centre offsets afterwards.
                features_ls[0] = torch.cat([f_center, features[:, :, 3:]], dim=-1)
            features_ls.append(f_center)
        if self._with_distance:
            features_ls.append(torch.norm(features[:, :, :3], 2, 2, keepdim=True))
        features = torch.cat(features_ls, dim=-1)
        mask = get_paddings_indicator(num_points, features.shape[1], axis=0)
        features = features * torch.unsqueeze(mask, -1).type_as(features)
        for pfn in self.pfn_layers:
            features = pfn(features, num_points)
        return features.squeeze(1)
