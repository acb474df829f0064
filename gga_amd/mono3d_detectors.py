"""Camera-only detector of the GGA retraining stage: ``FCOSMono3D`` (= ``SingleStageMono3DDetector``,
mmdet3d/models/detectors/single_stage_mono3d.py:16-78, fcos_mono3d.py:6-22) with the two third-party
pieces configs/_base_/models/fcos3d.py names - mmdet's ``ResNet`` (caffe-style bottlenecks, frozen
BatchNorm) and ``FPN`` - restated with mmdet's layer names so torchvision / detectron2 / mmdet
checkpoints load (parity unpinned: those modules are not in the reference tree).

Dense 2D convolutions: the 3x3 / stride-1 ones of the residual blocks qualify for ``gga_dense_conv3x3``
(channels-last activations, 64 .. 512 channels); the rest (7x7 stem, 1x1, strided) are MIOpen's.
"""
import torch
from torch import nn

from . import dense_conv
from .cnn import ConvModule, build_norm_layer
from .registry import BACKBONES, DETECTORS, NECKS, build_backbone, build_head, build_neck


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, style='pytorch', norm_cfg=dict(type='BN')):
        super().__init__()
        assert style in ('pytorch', 'caffe')
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)      # caffe: the stride sits on the first 1x1 conv
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=1, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = build_norm_layer(norm_cfg, planes * 4)[1]
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        # every convolution on the repo's kernels (1x1 and strided: gather-GEMM, 3x3 stride 1: dense kernel) and every
        # BatchNorm (+ residual) (+ ReLU) as one fused pass pair, training or norm_eval statistics alike
        from . import functional as F
        if self.downsample is None:
            identity = x
        else:
            identity = F.bn_act(dense_conv.conv2d(x, self.downsample[0]), self.downsample[1], relu=False)
        out = F.bn_act(dense_conv.conv2d(x, self.conv1), self.bn1, relu=True)
        out = F.bn_act(dense_conv.conv2d(out, self.conv2), self.bn2, relu=True)
        return F.bn_act(dense_conv.conv2d(out, self.conv3), self.bn3, relu=True, residual=identity)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, in_channels=3, stem_channels=64, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch', frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None, zero_init_residual=True, pretrained=None,
                 init_cfg=None, **kwargs):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet (bottleneck depths 50 / 101 / 152 are built)')
        assert dcn is None and set(dilations[:num_stages]) == {1}, 'plain ResNet as in configs/_base_/models/fcos3d.py'
        self.depth, self.num_stages, self.out_indices, self.frozen_stages, self.norm_eval = depth, num_stages, out_indices, frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(in_channels, stem_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, stem_channels)[1]
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = stem_channels
        self.res_layers = []
        for i, n in enumerate(self.arch_settings[depth][:num_stages]):
            planes = base_channels * 2 ** i
            blocks = []
            for j in range(n):
                stride = strides[i] if j == 0 else 1
                down = None
                if j == 0 and (stride != 1 or inplanes != planes * 4):
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                         build_norm_layer(norm_cfg, planes * 4)[1])
                blocks.append(Bottleneck(inplanes, planes, stride, down, style, norm_cfg))
                inplanes = planes * 4
            name = f'layer{i + 1}'
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        # mmdet's ResNet (un-vendored; restated from the published code): a string ``pretrained`` becomes
        # init_cfg = Pretrained; with no init_cfg the default is Kaiming on the convolutions, 1 on the norm weights and -
        # only in that case - zero_init_residual's 0 on every block's last norm weight. A Pretrained init_cfg replaces
        # all of that: nothing is zeroed, the checkpoint is loaded by init_weights()
        assert not (init_cfg and pretrained), 'init_cfg and pretrained cannot be specified at the same time'
        if isinstance(pretrained, str):
            init_cfg = dict(type='Pretrained', checkpoint=pretrained)
        self.init_cfg = init_cfg
        self.zero_init_residual = bool(zero_init_residual) and init_cfg is None
        self._default_init()
        self._freeze_stages()

    def _default_init(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def init_weights(self):
        """mmcv ``BaseModule.init_weights`` for this backbone: a ``Pretrained`` init_cfg loads its checkpoint (a local
        file, or a ``state_dict`` already in memory under ``init_cfg['state_dict']``); model-zoo URLs such as
        ``open-mmlab://detectron2/resnet101_caffe`` (configs/_base_/models/fcos3d.py:8-10) cannot be fetched without a
        network - that is said loudly and the Kaiming initialisation stays (NOT a zeroed residual branch)."""
        import os
        import warnings
        cfg = self.init_cfg
        if not cfg:
            return
        if cfg.get('type') != 'Pretrained':
            raise KeyError(f'init_cfg type {cfg.get("type")!r}: configs/gga only use Pretrained for the backbone')
        state = cfg.get('state_dict')
        ckpt = cfg.get('checkpoint')
        if state is None and isinstance(ckpt, str) and os.path.isfile(ckpt):
            state = torch.load(ckpt, map_location='cpu')
        if state is None:
            warnings.warn(f'ResNet: pretrained checkpoint {ckpt!r} is not a local file and cannot be downloaded here - the '
                          f'backbone keeps its random (Kaiming) initialisation; results will not match a run that loaded it')
            return
        state = state.get('state_dict', state)
        prefix = cfg.get('prefix')
        if prefix:
            state = {k[len(prefix):].lstrip('.'): v for k, v in state.items() if k.startswith(prefix)}
        missing, unexpected = self.load_state_dict(state, strict=False)
        if missing:
            warnings.warn(f'ResNet: {len(missing)} parameters not in the checkpoint, e.g. {missing[:3]}')

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode='nearest'), init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, list)
        self.in_channels, self.out_channels, self.num_ins, self.num_outs = in_channels, out_channels, len(in_channels), num_outs
        self.relu_before_extra_convs, self.upsample_cfg = relu_before_extra_convs, dict(upsample_cfg)
        self.backbone_end_level = self.num_ins if end_level in (-1, self.num_ins - 1) else end_level + 1
        assert num_outs >= self.backbone_end_level - start_level
        self.start_level = start_level
        assert isinstance(add_extra_convs, (str, bool))
        self.add_extra_convs = 'on_input' if add_extra_convs is True else add_extra_convs
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg if not no_norm_on_lateral else None, act_cfg=act_cfg, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                             act_cfg=act_cfg, inplace=False))
        extra = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra >= 1:
            for i in range(extra):
                cin = self.in_channels[self.backbone_end_level - 1] if i == 0 and self.add_extra_convs == 'on_input' else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                                 act_cfg=act_cfg, inplace=False))
        for m in self.modules():        # init_cfg: Xavier uniform on Conv2d
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        for i in range(len(laterals) - 1, 0, -1):
            laterals[i - 1] = laterals[i - 1] + nn.functional.interpolate(laterals[i], size=laterals[i - 1].shape[2:], **self.upsample_cfg)
        outs = [self.fpn_convs[i](laterals[i]) for i in range(len(laterals))]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for _ in range(self.num_outs - len(outs)):
                    outs.append(nn.functional.max_pool2d(outs[-1], 1, stride=2))
            else:
                src = {'on_input': inputs[self.backbone_end_level - 1], 'on_lateral': laterals[-1], 'on_output': outs[-1]}[self.add_extra_convs]
                outs.append(self.fpn_convs[len(laterals)](src))
                for i in range(len(laterals) + 1, self.num_outs):
                    outs.append(self.fpn_convs[i](torch.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]))
        return tuple(outs)


@DETECTORS.register_module()
class SingleStageMono3DDetector(nn.Module):
    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    with_neck = property(lambda self: getattr(self, 'neck', None) is not None)

    def init_weights(self):
        """What tools/train.py:222 calls on the built model (mmcv ``BaseModule.init_weights`` walks the children): the
        backbone's init_cfg (checkpoint), the head's normal(0.01) / focal-prior initialisation; the neck's Xavier
        initialisation is applied at construction."""
        for m in (self.backbone, getattr(self, 'neck', None), self.bbox_head):
            if m is not None and hasattr(m, 'init_weights'):
                m.init_weights()

    def extract_feat(self, img):
        x = self.backbone(img)
        return self.neck(x) if self.with_neck else x

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_3d, gt_labels_3d, centers2d, depths, attr_labels=None,
                      gt_bboxes_ignore=None):
        prepared = None
        if getattr(self.bbox_head, 'prepare_loss_before_forward', False):      # the target side of the loss first: its host waits end before the
            head = self.bbox_head                        # forward pass is queued (PGDHead.prepare_loss)
            prepared = head.prepare_loss(head.featmap_sizes_of(img.shape), img.shape[0], img.dtype, img.device, gt_bboxes, gt_labels,
                                         gt_bboxes_3d, gt_labels_3d, centers2d, depths, attr_labels)
        x = self.extract_feat(img)
        outs = self.bbox_head(x)        # BaseMono3DDenseHead.forward_train (base_mono3d_dense_head.py:17-71)
        kw = dict(prepared=prepared) if prepared is not None else {}
        return self.bbox_head.loss(*outs, gt_bboxes, gt_labels, gt_bboxes_3d, gt_labels_3d, centers2d, depths, attr_labels, img_metas,
                                   gt_bboxes_ignore=gt_bboxes_ignore, **kw)

    @torch.no_grad()
    def simple_test(self, img, img_metas, rescale=False):
        """single_stage_mono3d.py:80-114: ``img_bbox`` (3D boxes, scores, labels) and - with a 2D branch -
        ``img_bbox2d`` (per-class arrays [n, 5]) per image."""
        from .box3d import bbox3d2result
        outs = self.bbox_head(self.extract_feat(img))
        bbox_outputs = self.bbox_head.get_bboxes(*outs, img_metas, rescale=rescale)
        results = []
        for out in bbox_outputs:
            bboxes, scores, labels, attrs = out[:4]
            r = dict(img_bbox=bbox3d2result(bboxes, scores, labels, attrs))
            if self.bbox_head.pred_bbox2d:      # mmdet.core.bbox2result
                b2d, lab = out[4].cpu().numpy(), labels.cpu().numpy()
                r['img_bbox2d'] = [b2d[lab == i, :] for i in range(self.bbox_head.num_classes)]
            results.append(r)
        return results

    def forward_test(self, imgs, img_metas, **kwargs):
        if len(imgs) != 1:
            raise NotImplementedError('test-time augmentation is not built')
        return self.simple_test(imgs[0], img_metas[0], **kwargs)

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.forward_test(**kwargs)

    def _parse_losses(self, losses):
        from collections import OrderedDict
        log_vars = OrderedDict((k, v.mean() if isinstance(v, torch.Tensor) else sum(x.mean() for x in v)) for k, v in losses.items())
        loss = torch.stack([v for k, v in log_vars.items() if 'loss' in k]).sum()
        log_vars['loss'] = loss
        return loss, log_vars

    def train_step(self, data, optimizer=None):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))


@DETECTORS.register_module()
class FCOSMono3D(SingleStageMono3DDetector):
    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__(backbone, neck, bbox_head, train_cfg, test_cfg, pretrained)
