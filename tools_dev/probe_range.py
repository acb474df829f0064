import sys; sys.path.insert(0,'/root/repo')
import torch, math
from gga_amd import dense_conv
DEV='cuda:0'
torch.manual_seed(3)
conv = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(DEV)
conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
base = torch.randn(2, 64, 24, 40, device=DEV)
for span in (10, 20, 30, 40, 60, 100):
    e = torch.randint(-span, span+1, (2, 1, 24, 40), device=DEV).float()
    x = (base * torch.exp2(e)).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for planes in (3, 2):
            dense_conv.PLANES = planes
            y = dense_conv.conv2d(x, conv)
            mag = torch.nn.functional.conv2d(x.double().abs(), conv.weight.double().abs(), padding=1)
            ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), padding=1)
            sw = torch.nn.functional.conv2d(torch.ones_like(x).double(), conv.weight.double().abs(), padding=1)
            err = (y.double() - ref).abs()
            rel = float((err / mag).max())
            excess = (err - 2e-6 * mag).clamp(min=0) / (float(x.abs().max()) * sw)
            print(f'span 2^+-{span} planes {planes}: max err/mag {rel:.2e}; max (err - 2e-6 mag)/(absmax * sum|w|) = {float(excess.max()):.3e} = 2^{math.log2(max(float(excess.max()),1e-300)):.1f}; err/max|ref| {float(err.max()/ref.abs().max()):.2e}')
