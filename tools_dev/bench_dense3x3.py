"""Dense 3x3 bf16x9 kernel (gga_dense_conv3x3) against MIOpen's fp32 convolution: values and time
at the shapes of the BEV trunk / head branches. A 1 GiB memset runs between timed launches."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import _lib, functional as F
from gga_amd.sparse import _pack_weight

dev = 'cuda:0'
torch.backends.cudnn.benchmark = False


def run(B, C, Co, H, W):
    torch.manual_seed(0)
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(C, Co, 3, padding=1, bias=False).to(dev)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    from gga_amd import dense_conv
    out = {}
    def mine():
        out['y'] = dense_conv._run(x, conv.weight.detach(), False)[0]
    mine()
    with torch.no_grad():
        ref = conv(x)
    ref64 = torch.nn.functional.conv2d(x.double(), conv.weight.double(), padding=1)
    y = out['y']
    e_mine = float((y.double() - ref64).abs().max() / ref64.abs().max())
    e_ref = float((ref.double() - ref64).abs().max() / ref64.abs().max())
    trash = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    def timeit(fn, n=10):
        ts = []
        for i in range(n + 2):
            trash.fill_(float(i))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if i >= 2: ts.append(e0.elapsed_time(e1))
        return sum(ts) / len(ts)
    with torch.no_grad():
        t_ref = timeit(lambda: conv(x))
    t_mine = timeit(mine)
    gf = 2.0 * B * H * W * C * Co * 9 / 1e9
    if C % 64 == 0 and Co % 64 == 0:
        from gga_amd import dense_conv
        gy = torch.randn_like(y)
        gw = dense_conv._wgrad(x, gy, conv.weight)
        gw_ref = torch.ops.aten.convolution_backward(gy, x, conv.weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        gw64 = torch.nn.grad.conv2d_weight(x.double(), conv.weight.shape, gy.double(), padding=1)
        ew = float((gw.double() - gw64).abs().max() / gw64.abs().max()); ewr = float((gw_ref.double() - gw64).abs().max() / gw64.abs().max())
        tw = timeit(lambda: dense_conv._wgrad(x, gy, conv.weight))
        twr = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, conv.weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
        print(f'   wgrad: bf16x9 {tw*1e3:.0f} us ({gf/tw:.0f} TFLOP/s-eq, err {ew:.1e})   MIOpen {twr*1e3:.0f} us ({gf/twr:.0f} TFLOP/s, err {ewr:.1e})')
    print(f'[{B},{C}->{Co},{H},{W}] bf16x9 {t_mine*1e3:.0f} us ({gf/t_mine:.0f} TFLOP/s-eq, err {e_mine:.1e})   '
          f'MIOpen {t_ref*1e3:.0f} us ({gf/t_ref:.0f} TFLOP/s, err {e_ref:.1e})')


run(16, 64, 64, 248, 216)
run(2, 64, 64, 37, 45)
run(16, 128, 128, 124, 108)
run(16, 128, 64, 124, 108)
run(16, 256, 256, 62, 54)
