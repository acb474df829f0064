"""Head output convolutions as the step runs them: the 15 branches read their 64-channel column block of one
[B, H, W, 960] tensor one after another (3.3 GB: every call starts cold). Prints the average per call.
Run against experiment builds with tools_dev/run_with_lib.py (flags HM_ABL_*, HW_ABL_* of headconv.hip)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gga_amd import functional as F, _lib
L = _lib.lib()
dev = 'cuda:0'
B, H, W = 16, 248, 216
big = torch.randn(B, H, W, 960, device=dev)
ss = torch.cat([torch.rand(64, device=dev) + 0.5, torch.rand(64, device=dev) - 0.5])
tag = os.path.basename(_lib.LIB_PATH)


def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * 15) * 1e3


out = []
for cout in (1, 2, 3):
    w = torch.randn(cout, 64, 3, 3, device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    y = torch.empty(B, cout, H, W, device=dev)
    gy = torch.randn(B, cout, H, W, device=dev)
    dw = torch.empty_like(w); db = torch.empty_like(b)
    ws = torch.empty(L.gga_head_conv3x3_workspace_bytes(cout), dtype=torch.uint8, device=dev)

    def fwd():
        for k in range(15):
            L.gga_head_conv3x3_fwd(big.data_ptr() + 4 * 64 * k, 960, F._p(ss), F._p(w), F._p(b), B, H, W, 64, cout, F._p(y), F._stream())

    def wgrad():
        for k in range(15):
            L.gga_head_conv3x3_wgrad(big.data_ptr() + 4 * 64 * k, 960, F._p(ss), F._p(gy), B, H, W, 64, cout, F._p(dw), F._p(db),
                                     ws.data_ptr(), ws.numel(), F._stream())
    out.append('cout %d: fwd %5.1f us  wgrad(+final) %5.1f us' % (cout, timeit(fwd), timeit(wgrad)))
dense = [torch.randn(B, H, W, 64, device=dev) for _ in range(5)]
w = torch.randn(2, 64, 3, 3, device=dev) * 0.05; b = torch.zeros(2, device=dev); y = torch.empty(B, 2, H, W, device=dev)
def fwd_dense():
    for k in range(15):
        L.gga_head_conv3x3_fwd(F._p(dense[k % 5]), 64, F._p(ss), F._p(w), F._p(b), B, H, W, 64, 2, F._p(y), F._stream())
out.append('dense input cout 2: fwd %5.1f us' % timeit(fwd_dense))
print(tag, ' | '.join(out), flush=True)
