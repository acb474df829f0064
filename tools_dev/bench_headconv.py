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
    return e0.elapsed_time(e1) / n * 1e3
B, H, W = 16, 248, 216
for cout in (1, 3):
    conv = torch.nn.Conv2d(64, cout, 3, padding=1).to(dev)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(B, 64, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(B, cout, H, W, device=dev)
    with torch.no_grad():
        tf = timeit(lambda: F.head_conv3x3(x, conv)); te = timeit(lambda: conv(x))
    def fb():
        y = F.head_conv3x3(x, conv); y.backward(g)
    def eb():
        y = conv(x); y.backward(g)
    print(f'cout={cout}: fwd fused {tf:.0f} us eager {te:.0f} us | fwd+bwd fused {timeit(fb):.0f} us eager {timeit(eb):.0f} us')
import ctypes as C
from gga_amd import _lib
L = _lib.lib()
for cout in (1, 3):
    conv = torch.nn.Conv2d(64, cout, 3, padding=1).to(dev)
    x = torch.randn(B, 64, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, cout, H, W, device=dev)
    w = conv.weight.detach().contiguous()
    gw = torch.empty_like(w); gb = torch.empty(cout, device=dev)
    ws = torch.empty(L.gga_head_conv3x3_workspace_bytes(cout), dtype=torch.uint8, device=dev)
    t_wg = timeit(lambda: L.gga_head_conv3x3_wgrad(F._p(x), 64, None, F._p(g), B, H, W, 64, cout, F._p(gw), F._p(gb), F._p(ws), ws.numel(), F._stream()))
    t_gx = timeit(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    t_gw = timeit(lambda: torch.ops.aten.convolution_backward(g, x, w, [cout], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True]))
    print(f'cout={cout}: my wgrad {t_wg:.0f} us | aten grad_input only {t_gx:.0f} us | aten grad_weight+bias {t_gw:.0f} us')
