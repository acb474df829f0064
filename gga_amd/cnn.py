"""Layer builders the reference takes from the un-vendored ``mmcv.cnn``
(``ConvModule``, ``build_conv_layer``, ``build_norm_layer``,
``build_upsample_layer``; used at mmdet3d/models/backbones/second.py:4,
necks/second_fpn.py:4, dense_heads/centerpoint_head_gga.py:5). Only what the GGA
configs reach is provided; parameter / sub-module names follow mmcv so a reference
checkpoint's ``state_dict`` keys line up (``conv``, ``bn``, ``activate``).

The dense 2D convolutions themselves are MIOpen's (MFMA) — SURVEY.md §8 a4/a5 marks
them "not hand-written".
"""
import torch
from torch import nn

from .registry import CONV_LAYERS

CONV_LAYERS.register_module('Conv2d', module=nn.Conv2d)
CONV_LAYERS.register_module('Conv1d', module=nn.Conv1d)
CONV_LAYERS.register_module('Conv', module=nn.Conv2d)

_NORMS = {'BN': ('bn', nn.BatchNorm2d), 'BN1d': ('bn', nn.BatchNorm1d), 'BN2d': ('bn', nn.BatchNorm2d),
          'BN3d': ('bn', nn.BatchNorm3d)}
_UPSAMPLE = {'deconv': nn.ConvTranspose2d}


def build_conv_layer(cfg, *args, **kwargs):
    cfg = dict(type='Conv2d') if cfg is None else dict(cfg)
    layer_type = cfg.pop('type')
    cls = CONV_LAYERS.get(layer_type)
    if cls is None:
        raise KeyError(f'Unrecognized layer type {layer_type}')
    return cls(*args, **kwargs, **cfg)


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    layer_type = cfg.pop('type')
    if layer_type == 'GN':          # mmcv: abbreviation 'gn', GroupNorm(num_groups, num_channels)
        requires_grad = cfg.pop('requires_grad', True)
        cfg.setdefault('eps', 1e-5)
        layer = nn.GroupNorm(num_channels=num_features, **cfg)
        for p in layer.parameters():
            p.requires_grad = requires_grad
        return 'gn' + str(postfix), layer
    if layer_type not in _NORMS:
        raise KeyError(f'Unrecognized norm type {layer_type}')
    abbr, cls = _NORMS[layer_type]
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    layer = cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


def build_upsample_layer(cfg, *args, **kwargs):
    cfg = dict(cfg)
    layer_type = cfg.pop('type')
    if layer_type not in _UPSAMPLE:
        raise KeyError(f'Unrecognized upsample type {layer_type}')
    return _UPSAMPLE[layer_type](*args, **kwargs, **cfg)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0.0):
    if getattr(module, 'weight', None) is not None:
        nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0.0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


class ConvModule(nn.Module):
    """conv -> norm -> ReLU with mmcv's attribute names and 'auto' bias rule."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, order=('conv', 'norm', 'act')):
        super().__init__()
        assert order == ('conv', 'norm', 'act')
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        if hasattr(self.conv, 'init_weights') and not isinstance(self.conv, nn.Conv2d):
            pass                                 # DCNv2 initialises itself (mmcv ConvModule.init_weights skips such convs)
        else:
            kaiming_init(self.conv)
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.with_norm else None

    def forward(self, x):
        from . import dense_conv
        is_bn = self.with_norm and isinstance(self.norm, nn.modules.batchnorm._BatchNorm)
        x = dense_conv.conv2d(x, self.conv, bn_follows=is_bn and self.norm.training)
        return self.after_conv(x)

    def forward_levels(self, xs):
        """``[self(x) for x in xs]`` with the convolution of all maps in one launch where the kernel allows it (the
        weight-sharing towers of an FPN head: dense_conv.conv2d_levels); normalisation and activation map by map."""
        from . import dense_conv
        is_bn = self.with_norm and isinstance(self.norm, nn.modules.batchnorm._BatchNorm)
        if is_bn or type(self.conv) is not nn.Conv2d or not dense_conv.levels_eligible(self.conv, xs):
            return [self(x) for x in xs]
        return [self.after_conv(y) for y in dense_conv.conv2d_levels(xs, self.conv)]

    def after_conv(self, x):
        is_bn = self.with_norm and isinstance(self.norm, nn.modules.batchnorm._BatchNorm)
        if is_bn:
            from . import functional as F        # fused BN(+ReLU) HIP pass for channels-last activations
            return F.bn_act(x, self.norm, relu=self.with_activation)
        if self.with_norm and isinstance(self.norm, nn.GroupNorm):     # mono3d heads
            from . import functional as F
            if self.with_activation and not isinstance(self.activate, nn.ReLU):
                x = F.gn_act(x, self.norm, relu=False)
                return self.activate(x)
            return F.gn_act(x, self.norm, relu=self.with_activation)   # fused channels-last pass pair (eager fallback inside)
        if self.with_norm:
            x = self.norm(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def run_conv_bn_relu(seq, x):
    """Run an ``nn.Sequential`` made of (conv, BatchNorm, ReLU) triples — the block structure of
    SECOND / SECONDFPN — with the BatchNorm+ReLU pairs fused (``functional.bn_act``)."""
    from . import dense_conv
    from . import functional as F
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if (isinstance(m, nn.modules.batchnorm._BatchNorm) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)):
            x = F.bn_act(x, m, relu=True)
            i += 2
        else:
            if isinstance(m, nn.Conv2d):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                x = dense_conv.conv2d(x, m, bn_follows=isinstance(nxt, nn.modules.batchnorm._BatchNorm) and nxt.training)
            else:
                x = m(x)
            i += 1
    return x


def to_channels_last(model):
    """Put the dense 2D trunk (Conv2d / ConvTranspose2d weights) in channels-last memory so MIOpen
    runs its NHWC kernels; values, shapes and ``state_dict`` are unchanged. Sparse-conv weights
    (5-D) and everything else are left alone."""
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if type(m) is nn.Conv2d and m.out_channels <= 4 and m.in_channels == 64 and m.kernel_size == (3, 3):
                continue        # output convs of the head branches: gga_head_conv3x3_* read the weight as it is stored ([cout, 64, 3, 3]
                                # contiguous); in channels-last memory every call paid a layout copy (30 per step) and so did its gradient
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return model
