import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from gga_amd import dense_conv
torch.manual_seed(0)
cin, cout = 64, 64
w = torch.randn(cout, cin, 3, 3, device='cuda:0')
wp = dense_conv._pack(w, False).cpu().numpy().view(np.uint16)
wn = w.cpu().numpy()
def planes(v):
    u = np.float32(v).view(np.uint32); u1 = u & 0xFFFF0000
    r1 = np.float32(v) - u1.view(np.float32); u2 = r1.view(np.uint32) & 0xFFFF0000
    r2 = r1 - u2.view(np.float32)
    return u1 >> 16, u2 >> 16, r2.view(np.uint32) >> 16
CO = 64
bad = 0
for tap in range(9):
    for c in range(cin):
        for col in range(cout):
            p = planes(wn[col, c, tap // 3, tap % 3])
            st = tap * (cin // 16) + (c >> 4)
            for pl in range(3):
                got = wp[st * 3 * CO * 16 + pl * CO * 16 + col * 16 + (c & 15)]
                if got != p[pl]:
                    bad += 1
                    if bad < 5: print('mismatch', tap, c, col, pl, got, p[pl])
print('bad', bad, 'of', 9 * cin * cout * 3)
x = torch.randn(1, cin, 8, 32, device='cuda:0').contiguous(memory_format=torch.channels_last)
y = dense_conv._run(x, w, False)[0]
ref = torch.nn.functional.conv2d(x, w, padding=1)
print('conv err', float((y - ref).abs().max()), float(ref.abs().max()))
