import sys, torch
sys.path.insert(0, '.')
from gga_amd import functional as F
dev = 'cuda:0'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for shape in [(16, 64, 248, 216), (16, 128, 124, 108), (16, 256, 62, 54), (510000, 128)]:
    C = shape[1]
    bn = (torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d)(C, eps=1e-3, momentum=0.01).to(dev)
    x = torch.randn(*shape, device=dev)
    if len(shape) == 4: x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    g = torch.randn_like(x)
    mb = x.numel() * 4 / 1e6
    def fused():
        y = F.bn_act(x, bn, relu=True); y.backward(g)
    def eager():
        y = torch.relu(bn(x)); y.backward(g)
    def fused_f():
        with torch.no_grad(): pass
        return F.bn_act(x, bn, relu=True)
    def eager_f():
        return torch.relu(bn(x))
    tf, te, tff, tef = timeit(fused), timeit(eager), timeit(fused_f), timeit(eager_f)
    print(f'{shape}: {mb:.0f} MB  fused fwd+bwd {tf*1e3:.0f} us ({8*mb/tf/1e3:.2f} TB/s of 8 passes)  eager {te*1e3:.0f} us | fwd only fused {tff*1e3:.0f} us eager {tef*1e3:.0f} us')
