"""The stride-2 3x3 convolutions that open a ``SECOND`` stage (mmdet3d/models/backbones/second.py:49-57) and the
kernel = stride transposed convolutions of ``SECONDFPN`` (necks/second_fpn.py:52-69; stride 1 = a 1x1 product) on
the repo's own matrix kernels: forward, backward-data and weight gradient.

Over the pixel rows of a channels-last image all of them are gather-GEMMs ``y[row] = sum_k x[map[k][row]] W[k]``
with an ARITHMETIC rule book - exactly what the sparse-convolution kernels compute from a hashed one
(``gga_sparse_conv_apply_split`` / ``gga_sparse_conv_wgrad_split``: fp32 as six bf16 partial products,
deterministic weight gradient). The rule books depend on the shape only and are built once per
(batch, map size, kernel, stride, padding); rows are visited grouped by their set of valid taps, so a transposed
convolution's output rows run the one tap of their phase and border rows skip the padding taps.
Channel widths above 128 run as 128-wide column blocks (``*_strided`` entry points).
"""
import torch
from torch import nn

from . import _lib
from . import functional as F
from ._lib import check

ENABLED = True
_BOOKS = {}


class _Book:
    """Rule books of a k x k / stride s / padding p convolution [B, H, W] -> [B, Ho, Wo]:
    ``fwd`` [k*k, n_out]: input pixel of (tap, output pixel) or -1; ``bwd`` [k*k, n_in]: output pixel that
    reaches the input pixel through that tap or -1; row masks and mask-sorted row orders of both."""

    def __init__(self, B, H, W, k, s, p, device):
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        b = torch.arange(B, device=device).view(B, 1, 1)
        oy, ox = torch.arange(Ho, device=device).view(1, Ho, 1), torch.arange(Wo, device=device).view(1, 1, Wo)
        iy_, ix_ = torch.arange(H, device=device).view(1, H, 1), torch.arange(W, device=device).view(1, 1, W)
        fwd, bwd = [], []
        for ky in range(k):
            for kx in range(k):
                iy, ix = oy * s + ky - p, ox * s + kx - p
                ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
                fwd.append(torch.where(ok, b * (H * W) + iy * W + ix, -1).reshape(-1))
                ny, nx = iy_ + p - ky, ix_ + p - kx
                ok = (ny >= 0) & (nx >= 0) & (ny % s == 0) & (nx % s == 0) & (ny // s < Ho) & (nx // s < Wo)
                bwd.append(torch.where(ok, b * (Ho * Wo) + (ny // s) * Wo + nx // s, -1).reshape(-1))
        self.Ho, self.Wo, self.kvol = Ho, Wo, k * k
        self.n_in, self.n_out = B * H * W, B * Ho * Wo
        self.fwd, self.bwd = torch.stack(fwd).int().contiguous(), torch.stack(bwd).int().contiguous()
        self.fwd_mask, self.fwd_perm = self._order(self.fwd)
        self.bwd_mask, self.bwd_perm = self._order(self.bwd)

    @staticmethod
    def _order(m):
        kvol, n = m.shape
        if kvol > 32:
            return None, None
        mask = torch.empty(n, dtype=torch.int32, device=m.device)
        check(_lib.lib().gga_sparse_rowmask(F._p(m), n, kvol, F._p(mask), F._stream()), 'gga_sparse_rowmask')
        if bool((mask == mask[0]).all()):          # every row has the same taps: the natural order is the best one
            return mask, None
        from .sparse import mask_order
        return mask, mask_order(mask, kvol)


def book(B, H, W, k, s, p, device):
    key = (B, H, W, k, s, p, str(device))
    if key not in _BOOKS:
        with torch.no_grad():
            _BOOKS[key] = _Book(B, H, W, k, s, p, device)
    return _BOOKS[key]


def _rows(t):
    """[B, C, H, W] channels-last tensor as its [B*H*W, C] row matrix (no copy)."""
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C)


def _planes():
    from . import dense_conv
    return dense_conv.PLANES


def _amax(t):
    """Absmax bits of ``t`` (left by its producer when possible), None on the three-bf16-plane path."""
    from . import dense_conv
    return dense_conv.tensor_amax(t) if dense_conv.PLANES == 2 else None


def _apply(x_rows, m, mask, perm, w_kio, n_rows, x_amax=None, w_amax=None, want_stats=False, bank=False):
    """y [n_rows, cout] = sum_k x_rows[m[k]] @ w_kio[k]  (w_kio [kvol, cin, cout]). ``x_amax`` / ``w_amax``: absmax
    bits of the operands (two-plane arithmetic), computed here when missing. ``want_stats``: also return the f64
    [workgroups, 2, cout] per-channel sums of y for the BatchNorm that follows (``gga_sparse_conv_apply_stats``).
    ``bank``: ``w_kio`` is a VIEW ([kvol, cin, cout] or [k, k, cin, cout], any strides) of a parameter that lives across
    steps - its operand and absmax then come from the weight bank (refreshed with all others once per optimizer step)
    instead of a pack + absmax launch per call."""
    L = _lib.lib()
    kvol, cin, cout = (w_kio.shape[0] * w_kio.shape[1], w_kio.shape[2], w_kio.shape[3]) if w_kio.dim() == 4 else w_kio.shape
    planes = _planes()
    if planes == 2:
        x_amax = _amax(x_rows) if x_amax is None else x_amax
        if not bank:
            w_amax = _amax(w_kio) if w_amax is None else w_amax
        else:
            from . import dense_conv
            if dense_conv.RANGE_GUARD.armed:
                dense_conv.RANGE_GUARD.record(w_kio.detach())
    else:
        x_amax = w_amax = None
    y = torch.empty((n_rows, cout), dtype=torch.float32, device=x_rows.device)
    tiles = int(L.gga_sparse_conv_apply_tiles(n_rows)) if want_stats else 0
    parts = []
    for c0 in range(0, cout, 128):
        c1 = min(c0 + 128, cout)
        if bank:
            from . import weight_bank
            wp, w_amax = weight_bank.gather_operand(w_kio, planes, c0, c1)
        else:
            wp = torch.empty(L.gga_sparse_split_weight_bytes(kvol, cin, c1 - c0) // 2, dtype=torch.int16, device=y.device)
            check(L.gga_sparse_pack_weight_planes(F._p(w_kio[:, :, c0:c1].contiguous()), kvol, cin, c1 - c0, 0, planes, F._p(w_amax),
                                                  F._p(wp), F._stream()), 'gga_sparse_pack_weight_split')
        st = torch.empty((tiles, 2, c1 - c0), dtype=torch.float64, device=y.device) if want_stats else None
        check(L.gga_sparse_conv_apply_stats(F._p(x_rows), F._p(m), F._p(wp), F._p(perm), F._p(mask), n_rows, kvol, cin,
                                            c1 - c0, 0, y.data_ptr() + 4 * c0, cout, planes, F._p(x_amax), F._p(w_amax),
                                            F._p(st), F._stream()), 'gga_sparse_conv_apply_split_strided')
        if want_stats:
            parts.append(st)
    if want_stats:
        return y, (parts[0] if len(parts) == 1 else torch.cat(parts, dim=2))
    return y


def _wgrad(x_rows, g_rows, m, n_rows, x_amax=None, g_amax=None):
    """gw [kvol, cin, cout] = sum over rows of x_rows[m[k][row]]^T g_rows[row] (deterministic)."""
    planes = _planes()
    if planes == 2:
        x_amax = _amax(x_rows) if x_amax is None else x_amax
        g_amax = _amax(g_rows) if g_amax is None else g_amax
    else:
        x_amax = g_amax = None
    L = _lib.lib()
    kvol, cin, cout = m.shape[0], x_rows.shape[1], g_rows.shape[1]
    gw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=x_rows.device)
    for i0 in range(0, cin, 128):
        i1 = min(i0 + 128, cin)
        for o0 in range(0, cout, 128):
            o1 = min(o0 + 128, cout)
            whole = i1 - i0 == cin and o1 - o0 == cout
            part = gw if whole else torch.empty((kvol, i1 - i0, o1 - o0), dtype=torch.float32, device=gw.device)
            ws = F._workspace('sp_wgrad', L.gga_sparse_conv_wgrad_workspace_bytes(n_rows, kvol, i1 - i0, o1 - o0), gw.device)
            check(L.gga_sparse_conv_wgrad_planes(x_rows.data_ptr() + 4 * i0, cin, g_rows.data_ptr() + 4 * o0, cout, F._p(m),
                                                 n_rows, kvol, i1 - i0, o1 - o0, F._p(part), planes, F._p(x_amax), F._p(g_amax),
                                                 F._p(ws), ws.numel(), F._stream()), 'gga_sparse_conv_wgrad_split_strided')
            if not whole:
                gw[:, i0:i1, o0:o1] = part
    return gw


class _StridedConv(torch.autograd.Function):
    """Conv2d(k, stride s, padding p), weight [cout, cin, k, k]."""

    @staticmethod
    def forward(ctx, x, weight, k, s, p):
        B, cin, H, W = x.shape
        bk = book(B, H, W, k, s, p, x.device)
        w = weight.detach()
        x_amax = _amax(x)
        y, stats = _apply(_rows(x), bk.fwd, bk.fwd_mask, bk.fwd_perm, w.permute(2, 3, 1, 0), bk.n_out, x_amax, want_stats=True, bank=True)
        ctx.save_for_backward(x, weight)
        ctx.geom, ctx.amax = (k, s, p), (x_amax, None)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return y.view(B, bk.Ho, bk.Wo, -1).permute(0, 3, 1, 2), stats

    @staticmethod
    def backward(ctx, gy, _gstats=None):
        x, weight = ctx.saved_tensors
        k, s, p = ctx.geom
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        bk = book(B, H, W, k, s, p, x.device)
        gy = gy.contiguous(memory_format=torch.channels_last)
        g_rows = _rows(gy)
        gx = gw = None
        x_amax, w_amax = ctx.amax if _planes() == 2 else (None, None)
        g_amax = _amax(gy)
        if ctx.needs_input_grad[0]:
            w = weight.detach()
            gx = _apply(g_rows, bk.bwd, bk.bwd_mask, bk.bwd_perm, w.permute(2, 3, 0, 1), bk.n_in, g_amax, bank=True)
            gx = gx.view(B, H, W, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = _wgrad(_rows(x), g_rows, bk.fwd, bk.n_out, x_amax, g_amax).view(k, k, cin, cout).permute(3, 2, 0, 1)
        return gx, gw, None, None, None


class _Deconv(torch.autograd.Function):
    """ConvTranspose2d(kernel = stride = s, no padding), weight [cin, cout, s, s]: the rule books are those of the
    s x s / stride s convolution from the fine map to the coarse one, with the roles swapped."""

    @staticmethod
    def forward(ctx, x, weight, s):
        B, cin, H, W = x.shape
        bk = book(B, H * s, W * s, s, s, 0, x.device)          # fwd [s*s, n_coarse] fine pixel; bwd [s*s, n_fine] coarse pixel
        w = weight.detach()
        x_amax = _amax(x)
        y, stats = _apply(_rows(x), bk.bwd, bk.bwd_mask, bk.bwd_perm, w.permute(2, 3, 0, 1), bk.n_in, x_amax, want_stats=True, bank=True)
        ctx.save_for_backward(x, weight)
        ctx.s, ctx.amax = s, (x_amax, None)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return y.view(B, H * s, W * s, -1).permute(0, 3, 1, 2), stats

    @staticmethod
    def backward(ctx, gy, _gstats=None):
        x, weight = ctx.saved_tensors
        s = ctx.s
        B, cin, H, W = x.shape
        cout = weight.shape[1]
        bk = book(B, H * s, W * s, s, s, 0, x.device)
        gy = gy.contiguous(memory_format=torch.channels_last)
        g_rows = _rows(gy)
        gx = gw = None
        x_amax, w_amax = ctx.amax if _planes() == 2 else (None, None)
        g_amax = _amax(gy)
        if ctx.needs_input_grad[0]:
            w = weight.detach()
            gx = _apply(g_rows, bk.fwd, bk.fwd_mask, bk.fwd_perm, w.permute(2, 3, 1, 0), bk.n_out, g_amax, bank=True)
            gx = gx.view(B, H, W, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:      # [k][cout][cin] = sum over coarse rows of gy[fine(k, row)]^T x[row]
            gw = _wgrad(g_rows, _rows(x), bk.fwd, bk.n_out, g_amax, x_amax).view(s, s, cout, cin).permute(3, 2, 0, 1)
        return gx, gw, None


def _common(m, x):
    # widths above 256 would run as many 128 x 128 blocks that each re-read their operands (measured on ResNet-101's
    # 1x1 convolutions: 1282 weight-gradient launches, 153 ms per step against MIOpen's 7): those stay with the framework
    return (ENABLED and max(m.in_channels, m.out_channels) <= 256 and m.bias is None and m.groups == 1 and m.dilation == (1, 1) and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and m.in_channels % 4 == 0
            and m.out_channels % 4 == 0 and x.shape[0] * x.shape[2] * x.shape[3] * max(m.stride) ** 2 < 2 ** 31)


def eligible(m, x):
    if type(m) is nn.Conv2d:
        k, s, p = m.kernel_size, m.stride, m.padding
        return (_common(m, x) and m.padding_mode == 'zeros' and k[0] == k[1] and s[0] == s[1] and p[0] == p[1] and k[0] <= 5
                and (s[0] > 1 or k[0] == 1) and isinstance(p[0], int))
    if type(m) is nn.ConvTranspose2d:
        k, s = m.kernel_size, m.stride
        return (_common(m, x) and k == s and k[0] == k[1] and k[0] <= 5 and m.padding == (0, 0) and m.output_padding == (0, 0))
    return False


def conv(x, m):
    """``m(x)`` for an eligible Conv2d / ConvTranspose2d module; the result carries the per-channel sums of its values
    (``y.bn_partials``) that ``functional.bn_act`` uses instead of a reduce pass of its own."""
    if type(m) is nn.Conv2d:
        y, stats = _StridedConv.apply(x, m.weight, m.kernel_size[0], m.stride[0], m.padding[0])
    else:
        y, stats = _Deconv.apply(x, m.weight, m.stride[0])
    F.attach_bn_partials(y, stats)
    return y
