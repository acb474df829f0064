"""The stride-2 3x3 convolutions and the kernel=stride transposed convolutions of the SECOND / SECONDFPN
trunk as gather-GEMMs over pixel rows (the sparse-conv kernels with an arithmetic rule book) against the
framework's MIOpen kernels: time and error (both against float64) per pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gga_amd import _lib
from gga_amd import functional as F
from gga_amd.sparse import _pack_weight

DEV = 'cuda:0'
L = _lib.lib()


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def conv_maps(B, H, W, k, s, p):
    """fwd map [k*k, n_out] (input pixel of every (tap, output pixel) or -1) and its transpose [k*k, n_in]."""
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    b = torch.arange(B, device=DEV).view(B, 1, 1)
    oy, ox = torch.arange(Ho, device=DEV).view(1, Ho, 1), torch.arange(Wo, device=DEV).view(1, 1, Wo)
    iy_, ix_ = torch.arange(H, device=DEV).view(1, H, 1), torch.arange(W, device=DEV).view(1, 1, W)
    fwd, bwd = [], []
    for ky in range(k):
        for kx in range(k):
            iy, ix = oy * s + ky - p, ox * s + kx - p
            ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
            fwd.append(torch.where(ok, b * H * W + iy * W + ix, -1).reshape(-1))
            ny, nx = iy_ + p - ky, ix_ + p - kx
            ok = (ny % s == 0) & (nx % s == 0) & (ny // s < Ho) & (nx // s < Wo) & (ny >= 0) & (nx >= 0)
            bwd.append(torch.where(ok, b * Ho * Wo + (ny // s) * Wo + nx // s, -1).reshape(-1))
    return torch.stack(fwd).int().contiguous(), torch.stack(bwd).int().contiguous(), Ho, Wo


def mask_perm(m):
    kvol, n = m.shape
    mask = torch.empty(n, dtype=torch.int32, device=DEV)
    L.gga_sparse_rowmask(F._p(m), n, kvol, F._p(mask), F._stream())
    perm = torch.sort(mask, stable=True)[1].int()
    return mask, perm


def apply(x_rows, m, mask, perm, wp, n_rows, kvol, cin, cout):
    y = torch.empty((n_rows, cout), device=DEV)
    rc = L.gga_sparse_conv_apply_split(F._p(x_rows), F._p(m), F._p(wp), F._p(perm), F._p(mask), n_rows, kvol, cin, cout, 0, F._p(y),
                                       F._stream())
    _lib.check(rc, 'proto')
    return y


def wgrad(x_rows, g_rows, m, n_rows, kvol, cin, cout):
    gw = torch.empty((kvol, cin, cout), device=DEV)
    ws = F._workspace('proto_wgrad', L.gga_sparse_conv_wgrad_workspace_bytes(n_rows, kvol, cin, cout), DEV)
    rc = L.gga_sparse_conv_wgrad_split(F._p(x_rows), F._p(g_rows), F._p(m), n_rows, kvol, cin, cout, F._p(gw), F._p(ws), ws.numel(),
                                       F._stream())
    _lib.check(rc, 'proto')
    return gw


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def conv_case(B, cin, cout, H, W, k=3, s=2, p=1):
    torch.manual_seed(0)
    x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(cin, cout, k, s, p, bias=False).to(DEV).to(memory_format=torch.channels_last)
    fm, bm, Ho, Wo = conv_maps(B, H, W, k, s, p)
    n_out, n_in = B * Ho * Wo, B * H * W
    fmask, fperm = mask_perm(fm)
    bmask, bperm = mask_perm(bm)
    xr = x.permute(0, 2, 3, 1).reshape(n_in, cin)
    w = conv.weight.detach()
    wf = _pack_weight(w.permute(2, 3, 1, 0).contiguous(), k * k, cin, cout, 0, split=True)
    wb = _pack_weight(w.permute(2, 3, 0, 1).contiguous(), k * k, cout, cin, 0, split=True)
    y = apply(xr, fm, fmask, fperm, wf, n_out, k * k, cin, cout)
    gy = torch.randn(B, cout, Ho, Wo, device=DEV).contiguous(memory_format=torch.channels_last)
    gr = gy.permute(0, 2, 3, 1).reshape(n_out, cout)
    gx = apply(gr, bm, bmask, bperm, wb, n_in, k * k, cout, cin)
    gw = wgrad(xr, gr, fm, n_out, k * k, cin, cout)
    # float64 truth on a slice of the batch (the full map in float64 is slow on the device too)
    with torch.no_grad():
        y64 = torch.nn.functional.conv2d(x[:2].double(), w.double(), None, s, p)
        gx64, gw64 = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [s, s], [p, p], [1, 1], False,
                                                         [0, 0], 1, [True, True, False])[:2]
        y32 = conv(x)
        gx32, gw32 = torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, True, False])[:2]
    print(f'conv {cin}->{cout} k{k} s{s} on {B}x{H}x{W}:')
    print(f'   fwd    mine {timeit(lambda: apply(xr, fm, fmask, fperm, wf, n_out, k * k, cin, cout)):7.0f} us  err {rel(y.view(B, Ho, Wo, cout)[:2].permute(0, 3, 1, 2), y64):.1e}'
          f' | MIOpen {timeit(lambda: conv(x)):7.0f} us err {rel(y32[:2], y64):.1e}')
    t_m = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, False, False]))
    print(f'   bwd-d  mine {timeit(lambda: apply(gr, bm, bmask, bperm, wb, n_in, k * k, cout, cin)):7.0f} us  err {rel(gx.view(B, H, W, cin).permute(0, 3, 1, 2), gx64):.1e}'
          f' | MIOpen {t_m:7.0f} us err {rel(gx32, gx64):.1e}')
    t_m = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [False, True, False]))
    print(f'   wgrad  mine {timeit(lambda: wgrad(xr, gr, fm, n_out, k * k, cin, cout)):7.0f} us  err {rel(gw.view(k, k, cin, cout).permute(3, 2, 0, 1), gw64):.1e}'
          f' | MIOpen {t_m:7.0f} us err {rel(gw32, gw64):.1e}')


def deconv_case(B, cin, cout, H, W, s):
    """ConvTranspose2d(cin, cout, s, stride=s): its forward is the backward-data form of a k=s stride-s conv
    cout -> cin, so the maps are those of that conv with the roles swapped."""
    torch.manual_seed(0)
    x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    dec = torch.nn.ConvTranspose2d(cin, cout, s, s, bias=False).to(DEV).to(memory_format=torch.channels_last)
    Ho, Wo = H * s, W * s
    fm, bm, _, _ = conv_maps(B, Ho, Wo, s, s, 0)          # fm: [s*s, n_lo] hi pixel of (phase, lo pixel); bm: [s*s, n_hi]
    n_lo, n_hi = B * H * W, B * Ho * Wo
    bmask, bperm = mask_perm(bm)
    fmask, fperm = mask_perm(fm)
    w = dec.weight.detach()                                # [cin, cout, s, s]
    wf = _pack_weight(w.permute(2, 3, 0, 1).contiguous(), s * s, cin, cout, 0, split=True)
    wb = _pack_weight(w.permute(2, 3, 1, 0).contiguous(), s * s, cout, cin, 0, split=True)
    xr = x.permute(0, 2, 3, 1).reshape(n_lo, cin)
    y = apply(xr, bm, bmask, bperm, wf, n_hi, s * s, cin, cout)
    gy = torch.randn(B, cout, Ho, Wo, device=DEV).contiguous(memory_format=torch.channels_last)
    gr = gy.permute(0, 2, 3, 1).reshape(n_hi, cout)
    gx = apply(gr, fm, fmask, fperm, wb, n_lo, s * s, cout, cin)
    gw = wgrad(gr, xr, fm, n_lo, s * s, cout, cin)          # [k][cout][cin]
    with torch.no_grad():
        y64 = torch.nn.functional.conv_transpose2d(x[:2].double(), w.double(), None, s)
        y32 = dec(x)
        args = (None, [s, s], [0, 0], [1, 1], True, [0, 0], 1)
        gx64, gw64 = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), *args, [True, True, False])[:2]
        gx32, gw32 = torch.ops.aten.convolution_backward(gy, x, w, *args, [True, True, False])[:2]
    print(f'deconv {cin}->{cout} k{s} s{s} on {B}x{H}x{W}:')
    print(f'   fwd    mine {timeit(lambda: apply(xr, bm, bmask, bperm, wf, n_hi, s * s, cin, cout)):7.0f} us  err {rel(y.view(B, Ho, Wo, cout)[:2].permute(0, 3, 1, 2), y64):.1e}'
          f' | MIOpen {timeit(lambda: dec(x)):7.0f} us err {rel(y32[:2], y64):.1e}')
    t_m = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, *args, [True, False, False]))
    print(f'   bwd-d  mine {timeit(lambda: apply(gr, fm, fmask, fperm, wb, n_lo, s * s, cout, cin)):7.0f} us  err {rel(gx.view(B, H, W, cin).permute(0, 3, 1, 2), gx64):.1e}'
          f' | MIOpen {t_m:7.0f} us err {rel(gx32, gx64):.1e}')
    t_m = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, *args, [False, True, False]))
    print(f'   wgrad  mine {timeit(lambda: wgrad(gr, xr, fm, n_lo, s * s, cout, cin)):7.0f} us  err {rel(gw.view(s, s, cout, cin).permute(3, 2, 0, 1), gw64):.1e}'
          f' | MIOpen {t_m:7.0f} us err {rel(gw32, gw64):.1e}')


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    conv_case(B, 64, 128, 248, 216)
    deconv_case(B, 128, 128, 124, 108, 2)
    deconv_case(B, 64, 128, 248, 216, 1)
