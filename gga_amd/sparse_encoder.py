"""``SparseEncoder`` (SECOND middle encoder), ``SparseBasicBlock`` and
``make_sparse_convmodule`` — constructor arguments, layer structure and ``state_dict`` names
of the reference (mmdet3d/models/middle_encoders/sparse_encoder.py:43-214,
mmdet3d/ops/sparse_block.py:82-199; ``BasicBlock`` attribute names conv1/bn1/conv2/bn2 as in
mmdet's resnet), on the sparse layers of ``gga_amd.sparse``."""
import os
import torch
from torch import nn

from . import functional as F
from .cnn import build_conv_layer, build_norm_layer
from .registry import MIDDLE_ENCODERS
from .sparse import SparseConvTensor, SparseModule, SparseSequential


class SparseBasicBlock(SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None):
        super().__init__()
        assert downsample is None and stride == 1
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=1, dilation=1, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    norm1 = property(lambda self: getattr(self, self.norm1_name))
    norm2 = property(lambda self: getattr(self, self.norm2_name))

    def forward(self, x):
        identity = x.features
        assert x.features.dim() == 2, f'x.features.dim()={x.features.dim()}'
        out = self.conv1(x)
        out = out.replace_feature(F.bn_act(out.features, self.norm1, relu=True))
        out = self.conv2(out)
        # relu(norm2(.) + identity) in one fused pass
        out = out.replace_feature(F.bn_act(out.features, self.norm2, relu=True, residual=identity))
        return out


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type='SubMConv3d', norm_cfg=None, order=('conv', 'norm', 'act')):
    assert isinstance(order, tuple) and len(order) <= 3
    assert set(order) | {'conv', 'norm', 'act'} == {'conv', 'norm', 'act'}
    conv_cfg = dict(type=conv_type, indice_key=indice_key)
    layers = []
    for layer in order:
        if layer == 'conv':
            layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                           padding=padding, bias=False))
        elif layer == 'norm':
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == 'act':
            layers.append(nn.ReLU(inplace=True))
    return SparseSequential(*layers)


# GGA_SPARSE_DIRECT_BEV=0: the NCHW scatter + layout copy of rounds 1-2 (A/B switch)
DIRECT_BEV = os.environ.get('GGA_SPARSE_DIRECT_BEV', '1') == '1'


@MIDDLE_ENCODERS.register_module()
class SparseEncoder(nn.Module):
    def __init__(self, in_channels, sparse_shape, order=('conv', 'norm', 'act'),
                 norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16, ), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1, ), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type='conv_module',
                 channels_last=False):
        super().__init__()
        # extension (not a reference key): hand the BEV map to the 2D trunk in channels-last memory - the same logical
        # [N, C*D, H, W] tensor - so that its first convolution runs on the repo's kernels as well
        self.channels_last = channels_last
        assert block_type in ['conv_module', 'basicblock']
        self.sparse_shape = sparse_shape
        self.in_channels = in_channels
        self.order = order
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.stage_num = len(self.encoder_channels)
        self.fp16_enabled = False
        assert isinstance(order, tuple) and len(order) == 3 and set(order) == {'conv', 'norm', 'act'}
        if self.order[0] != 'conv':      # pre-activate
            self.conv_input = make_sparse_convmodule(in_channels, self.base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key='subm1', conv_type='SubMConv3d', order=('conv', ))
        else:
            self.conv_input = make_sparse_convmodule(in_channels, self.base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key='subm1', conv_type='SubMConv3d')
        encoder_out_channels = self.make_encoder_layers(make_sparse_convmodule, norm_cfg, self.base_channels,
                                                        block_type=block_type)
        self.conv_out = make_sparse_convmodule(encoder_out_channels, self.output_channels, kernel_size=(3, 1, 1),
                                               stride=(2, 1, 1), norm_cfg=norm_cfg, padding=0,
                                               indice_key='spconv_down2', conv_type='SparseConv3d')

    def build_indices(self, coors, batch_size):
        """Levels and rule books of this encoder for ``coors`` (see ``sparse.build_index_plan``); the plan
        travels with the coordinates (``coors.index_plan``) and ``forward`` picks it up."""
        from .sparse import build_index_plan
        coors = coors if coors.dtype == torch.int32 else coors.int()
        plan = build_index_plan(self, coors, self.sparse_shape, int(batch_size))
        # The plan's first level keeps `coors` itself; hung on that same tensor object it would close a reference cycle
        # (tensor -> plan -> level -> tensor) that only the cyclic collector frees - a generation-2 pass every ~12 steps, until
        # which every step's 1.6 GB of levels and rule books stayed allocated and the caching allocator kept calling hipMalloc
        # (tools_dev/who_holds.py; the "first-process transient" of the sparse leg). The plan travels on a second tensor object
        # over the same memory instead, which dies by reference counting with the step's inputs.
        carrier = coors.detach()
        carrier.__dict__.update(coors.__dict__)          # (the voxelizer's device-side count: coors.num_valid)
        carrier.index_plan = plan
        return carrier

    def forward(self, voxel_features, coors, batch_size):
        plan = getattr(coors, 'index_plan', None)
        if plan is not None and plan.level0.n == voxel_features.shape[0]:
            x = SparseConvTensor(voxel_features, plan.level0.coors, self.sparse_shape, int(batch_size), _level=plan.level0)
            x.indice_dict = plan.indice_dict
        else:
            x = SparseConvTensor(voxel_features, coors.int(), self.sparse_shape, int(batch_size))
        x = self.conv_input(x)
        encode_features = []
        for encoder_layer in self.encoder_layers:
            x = encoder_layer(x)
            encode_features.append(x)
        out = self.conv_out(encode_features[-1])
        f = out.features
        if (self.channels_last and DIRECT_BEV and f.is_cuda and f.dtype == torch.float32 and f.shape[1] % 4 == 0 and f.shape[0] > 0
                and out.indices.dtype == torch.int32):
            # the map straight into channels-last memory (one memset + one pass over the sites; the NCHW scatter, the layout
            # copy and their backward counterparts were 1.2 ms of the 58 ms step)
            from . import functional as F, dense_conv
            D, H, W = out.spatial_shape
            spatial_features = F.sparse_bev_channels_last(f, out.indices, out.batch_size, D, H, W)
            c = getattr(f, '_gga_amax', None)           # (version, pointer, size, absmax slot, pool generation) left by f's producer
            if (c is not None and dense_conv.PLANES == 2 and c[0] == f._version and c[1] == f.data_ptr() and c[2] == f.numel()
                    and c[4] == dense_conv.AMAX_POOL.generation):
                dense_conv.set_amax(spatial_features, c[3])     # the same values plus zeros: the features' absmax is the map's
            return spatial_features
        spatial_features = out.dense()
        N, C, D, H, W = spatial_features.shape
        spatial_features = spatial_features.view(N, C * D, H, W)
        if self.channels_last:
            spatial_features = spatial_features.contiguous(memory_format=torch.channels_last)
        return spatial_features

    def make_encoder_layers(self, make_block, norm_cfg, in_channels, block_type='conv_module',
                            conv_cfg=dict(type='SubMConv3d')):
        assert block_type in ['conv_module', 'basicblock']
        self.encoder_layers = SparseSequential()
        for i, blocks in enumerate(self.encoder_channels):
            blocks_list = []
            for j, out_channels in enumerate(tuple(blocks)):
                padding = tuple(self.encoder_paddings[i])[j]
                last_of_stage = j == len(blocks) - 1 and i != len(self.encoder_channels) - 1
                if i != 0 and j == 0 and block_type == 'conv_module':
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                  padding=padding, indice_key=f'spconv{i + 1}', conv_type='SparseConv3d'))
                elif block_type == 'basicblock':
                    if last_of_stage:
                        blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                      padding=padding, indice_key=f'spconv{i + 1}',
                                                      conv_type='SparseConv3d'))
                    else:
                        blocks_list.append(SparseBasicBlock(out_channels, out_channels, norm_cfg=norm_cfg,
                                                            conv_cfg=conv_cfg))
                else:
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, padding=padding,
                                                  indice_key=f'subm{i + 1}', conv_type='SubMConv3d'))
                in_channels = out_channels
            self.encoder_layers.add_module(f'encoder_layer{i + 1}', SparseSequential(*blocks_list))
        return out_channels
