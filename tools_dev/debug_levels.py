import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, mono3d_heads, dense_conv
from gga_amd.cnn import to_channels_last, ConvModule
import bench
dev = torch.device('cuda:0')
cfg = Config.fromfile(bench.PGD_CONFIG)
model = to_channels_last(build_model(cfg.model).to(dev)); model.train()
head = model.bbox_head
B = 12
sizes = head.featmap_sizes_of((B, 3, 384, 1248))
feats = [torch.randn(B, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for h, w in sizes]
n = {'lv': 0, 'single': 0}
o1, o2 = dense_conv._Conv3x3Levels.forward, dense_conv._Conv3x3.forward
def f1(ctx, w, *xs): n['lv'] += 1; return o1(ctx, w, *xs)
def f2(ctx, x, w, s): n['single'] += 1; return o2(ctx, x, w, s)
dense_conv._Conv3x3Levels.forward = staticmethod(f1); dense_conv._Conv3x3.forward = staticmethod(f2)
outs = head(feats)
print(n)
for name, m in head.named_modules():
    if isinstance(m, ConvModule):
        print(name, type(m.conv).__name__, m.conv.in_channels, m.conv.out_channels, 'bias' if m.conv.bias is not None else 'nobias', type(m.norm).__name__ if m.with_norm else None, dense_conv.levels_eligible(m.conv, feats) if type(m.conv) is torch.nn.Conv2d and m.conv.in_channels == 256 else '-')
