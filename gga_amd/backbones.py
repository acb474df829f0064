"""``SECOND`` backbone and ``SECONDFPN`` neck (reference mmdet3d/models/backbones/
second.py:24-91, necks/second_fpn.py:26-91): every convolution on the repo's own matrix kernels
(dense_conv.py: 3x3 stride 1; strided_conv.py: stride-2 3x3 and the transposed convolutions; pillar_conv.py: the first one)."""
import numpy as np
import torch
from torch import nn

from . import pillar_conv
from .cnn import build_conv_layer, build_norm_layer, build_upsample_layer, kaiming_init, run_conv_bn_relu
from .registry import BACKBONES, NECKS


@BACKBONES.register_module()
class SECOND(nn.Module):
    def __init__(self, in_channels=128, out_channels=[128, 128, 256], layer_nums=[3, 5, 5],
                 layer_strides=[2, 2, 2], norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01),
                 conv_cfg=dict(type='Conv2d', bias=False), init_cfg=None, pretrained=None):
        super().__init__()
        assert len(layer_strides) == len(layer_nums) == len(out_channels)
        in_filters = [in_channels, *out_channels[:-1]]
        blocks = []
        for i, layer_num in enumerate(layer_nums):
            block = [build_conv_layer(conv_cfg, in_filters[i], out_channels[i], 3, stride=layer_strides[i], padding=1),
                     build_norm_layer(norm_cfg, out_channels[i])[1], nn.ReLU(inplace=True)]
            for _ in range(layer_num):
                block += [build_conv_layer(conv_cfg, out_channels[i], out_channels[i], 3, padding=1),
                          build_norm_layer(norm_cfg, out_channels[i])[1], nn.ReLU(inplace=True)]
            blocks.append(nn.Sequential(*block))
        self.blocks = nn.ModuleList(blocks)
        self.pillar_backward = True     # see pillar_conv.py; False = dense backward through the canvas
        self.init_weights()

    def init_weights(self):
        for m in self.modules():          # init_cfg = dict(type='Kaiming', layer='Conv2d')
            if isinstance(m, nn.Conv2d):
                kaiming_init(m)

    def forward(self, x):
        outs = []
        for i, blk in enumerate(self.blocks):
            if i == 0 and self.pillar_backward and pillar_conv.eligible(blk[0], x):
                # same forward; the backward of this convolution runs on the occupied cells only
                x = run_conv_bn_relu(list(blk)[1:], pillar_conv.pillar_conv2d(x, blk[0]))
            else:
                x = run_conv_bn_relu(blk, x)
            outs.append(x)
        return tuple(outs)


@NECKS.register_module()
class SECONDFPN(nn.Module):
    def __init__(self, in_channels=[128, 128, 256], out_channels=[256, 256, 256], upsample_strides=[1, 2, 4],
                 norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01), upsample_cfg=dict(type='deconv', bias=False),
                 conv_cfg=dict(type='Conv2d', bias=False), use_conv_for_no_stride=False, init_cfg=None):
        super().__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.fp16_enabled = False
        deblocks = []
        for i, out_channel in enumerate(out_channels):
            stride = upsample_strides[i]
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                up = build_upsample_layer(upsample_cfg, in_channels=in_channels[i], out_channels=out_channel,
                                          kernel_size=upsample_strides[i], stride=upsample_strides[i])
            else:
                stride = int(np.round(1 / stride))
                up = build_conv_layer(conv_cfg, in_channels=in_channels[i], out_channels=out_channel,
                                      kernel_size=stride, stride=stride)
            deblocks.append(nn.Sequential(up, build_norm_layer(norm_cfg, out_channel)[1], nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(deblocks)
        for m in self.modules():          # init_cfg: Kaiming on ConvTranspose2d
            if isinstance(m, nn.ConvTranspose2d):
                kaiming_init(m)

    def forward(self, x):
        assert len(x) == len(self.in_channels)
        if all(len(d) == 3 and isinstance(d[1], nn.modules.batchnorm._BatchNorm) and isinstance(d[2], nn.ReLU)
               for d in self.deblocks):
            # every branch normalises straight into its channel slice of the concatenated map
            from . import functional as F
            from . import dense_conv
            return [F.bn_relu_cat([dense_conv.conv2d(x[i], d[0]) for i, d in enumerate(self.deblocks)], [d[1] for d in self.deblocks])]
        ups = [run_conv_bn_relu(deblock, x[i]) for i, deblock in enumerate(self.deblocks)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]
