"""Head output conv (forward, weight gradient) on maps with several tiles per workgroup and a column-block input, against float64."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gga_amd import functional as F, _lib
L = _lib.lib()
dev = 'cuda:0'
torch.manual_seed(0)
for (B, H, W, wide, blk) in ((6, 200, 176, 192, 1), (16, 248, 216, 128, 1), (2, 200, 176, 64, 0), (3, 37, 29, 64, 0)):
    big = torch.randn(B, H, W, wide, device=dev)
    xs = big[..., 64 * blk:64 * blk + 64]
    for cout in (1, 2, 3):
        for aff in (True, False):
            ss = torch.cat([torch.rand(64, device=dev) + 0.5, torch.rand(64, device=dev) - 0.5])
            w = torch.randn(cout, 64, 3, 3, device=dev) * 0.05
            b = torch.randn(cout, device=dev)
            y = torch.empty(B, cout, H, W, device=dev)
            L.gga_head_conv3x3_fwd(big.data_ptr() + 4 * 64 * blk, wide, F._p(ss) if aff else None, F._p(w), F._p(b), B, H, W, 64, cout, F._p(y), F._stream())
            xin = xs.permute(0, 3, 1, 2).double()
            if aff:
                xin = torch.relu(xin * ss[:64].double().view(1, -1, 1, 1) + ss[64:].double().view(1, -1, 1, 1))
            ref = torch.nn.functional.conv2d(xin, w.double(), b.double(), padding=1)
            e_f = float((y.double() - ref).abs().max() / ref.abs().max())
            gy = torch.randn(B, cout, H, W, device=dev)
            dw = torch.empty_like(w); db = torch.empty_like(b)
            ws = torch.empty(L.gga_head_conv3x3_workspace_bytes(cout), dtype=torch.uint8, device=dev)
            L.gga_head_conv3x3_wgrad(big.data_ptr() + 4 * 64 * blk, wide, F._p(ss) if aff else None, F._p(gy), B, H, W, 64, cout, F._p(dw), F._p(db), ws.data_ptr(), ws.numel(), F._stream())
            xr = xin.clone().requires_grad_(False)
            wr = w.double().clone().requires_grad_(True); br = b.double().clone().requires_grad_(True)
            torch.nn.functional.conv2d(xr, wr, br, padding=1).backward(gy.double())
            e_w = float((dw.double() - wr.grad).abs().max() / wr.grad.abs().max())
            e_b = float((db.double() - br.grad).abs().max() / br.grad.abs().max())
            flag = '' if max(e_f, e_w, e_b) < 1e-5 else '   <<<<<<'
            print((B, H, W, wide), 'cout', cout, 'aff', aff, 'fwd %.1e  dw %.1e  db %.1e' % (e_f, e_w, e_b), flag, flush=True)
