"""Forward time of gga_dense_conv3x3 at the two dominant shapes (1 GiB fill between launches)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import dense_conv
dev = 'cuda:0'
trash = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for (B, C, Co, H, W) in ((16, 64, 64, 248, 216), (16, 128, 128, 124, 108)):
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    ts = []
    for i in range(12):
        trash.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        wp = dense_conv._pack(w, False, dense_conv._transposed(H, W))
        e0.record(); dense_conv._run(x, w, False); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1))
    t = sum(ts) / len(ts)
    print(f'[{B},{C}->{Co},{H},{W}] {t*1e3:.0f} us  {2.0*B*H*W*C*Co*9/t/1e9:.0f} TF/s-eq (includes the 5 us weight pack)')
