"""``ModulatedDeformConv2dPack`` ('DCNv2' in ``CONV_LAYERS``) - the deformable convolution the PGD /
FCOS3D heads put at the end of their towers (``dcn_on_last_conv=True``,
mmdet3d/models/dense_heads/anchor_free_mono3d_head.py:187-211; configs/_base_/models/pgd.py:47;
configs/gga/gga_pdg.py inherits it). The op itself is mmcv's (third-party): constructor arguments,
parameter names (``weight``, ``bias``, ``conv_offset.*``), the zero-initialised offset branch and the
``chunk -> cat(o1, o2) / sigmoid(mask)`` wiring follow mmcv's published module so reference
checkpoints load; the arithmetic is the published DCNv2 definition (oracle/dcn_ref.py restates it in
plain torch).

Sampling (``gga_dcn_im2col``) and its backward (``gga_dcn_col2im``) are hand-written HIP kernels on
channels-last activations; the two GEMMs are library GEMMs (``torch.matmul`` -> hipBLASLt).
"""
import math

import torch
from torch import nn

from . import _lib
from . import functional as F
from ._lib import check
from .registry import CONV_LAYERS


def _pair(v):
    return (int(v), int(v)) if not isinstance(v, (tuple, list)) else (int(v[0]), int(v[1]))


MATRIX_KERNELS = True     # the three products of a DCN call on the repo's gather-GEMM kernels instead of the library's GEMM
_MAPS = {}


def _maps(n_pix, K, device):
    """Rule books that make the three matrix products of a DCN call gather-GEMMs over the rows of the column matrix
    viewed as [n_pix * K, C] (``strided_conv._apply`` / ``_wgrad``): ``fwd`` [K, n_pix]: row of (tap, pixel);
    ``bwd`` [K, n_pix * K]: the pixel whose gradient reaches column row r through tap r % K (-1 for the other taps),
    with its row masks / mask-sorted order. Built once per (pixels, taps)."""
    key = (n_pix, K, str(device))
    if key not in _MAPS:
        from .strided_conv import _Book
        with torch.no_grad():
            p = torch.arange(n_pix, device=device, dtype=torch.int32)
            fwd = torch.stack([p * K + k for k in range(K)]).contiguous()
            r = torch.arange(n_pix * K, device=device, dtype=torch.int32)
            bwd = torch.stack([torch.where(r % K == k, r // K, -1) for k in range(K)]).int().contiguous()
            _MAPS[key] = (fwd, *_Book._order(fwd), bwd, *_Book._order(bwd))
    return _MAPS[key]


def _use_matrix_kernels(C, cout, n_pix, K):
    return MATRIX_KERNELS and C % 32 == 0 and cout % 32 == 0 and C <= 256 and cout <= 256 and n_pix * K < 2 ** 31


class _ModulatedDeformConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, padding, dilation):
        F._need_cuda(x, offset, mask, weight)
        B, C, H, W = x.shape
        cout, _, kh, kw = weight.shape
        xl = x.contiguous(memory_format=torch.channels_last)          # [B,H,W,C] in memory
        offset, mask = offset.contiguous().float(), mask.contiguous().float()
        Ho, Wo = offset.shape[2], offset.shape[3]
        L = _lib.lib()
        n_pix, K = B * Ho * Wo, kh * kw
        col = torch.empty((n_pix, K * C), dtype=torch.float32, device=x.device)
        geom = (B, H, W, C, kh, kw, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1])
        ctx.matrix = _use_matrix_kernels(C, cout, n_pix, K)
        col_amax = None
        if ctx.matrix:
            # y[p] = sum_k col[p * K + k] W_k: the gather-GEMM of the strided convolutions over the rows of col (two fp16
            # planes on the matrix cores; the library's fp32 GEMM ran these shapes at 57 TFLOP/s). im2col leaves max |col|.
            from . import dense_conv, strided_conv
            col_amax = dense_conv.new_amax(x.device)
            check(L.gga_dcn_im2col_amax(F._p(xl), F._p(offset), F._p(mask), *geom, F._p(col), F._p(col_amax), F._stream()),
                  'gga_dcn_im2col')
            fwd, fmask, fperm = _maps(n_pix, K, x.device)[:3]
            w_kio = weight.detach().permute(2, 3, 1, 0).reshape(K, C, cout).contiguous()
            y = strided_conv._apply(col.view(n_pix * K, C), fwd, fmask, fperm, w_kio, n_pix, col_amax)
        else:
            check(L.gga_dcn_im2col(F._p(xl), F._p(offset), F._p(mask), *geom, F._p(col), F._stream()), 'gga_dcn_im2col')
            wmat = weight.permute(0, 2, 3, 1).reshape(cout, kh * kw * C)       # [Cout, tap, C]: the column order
            y = col @ wmat.t()
        if bias is not None:
            y = y + bias
        ctx.save_for_backward(xl, offset, mask, weight, col if ctx_keep_col(col) else None)
        ctx.col_amax = col_amax if ctx_keep_col(col) else None
        ctx.geom, ctx.has_bias, ctx.out_hw = geom, bias is not None, (Ho, Wo)
        # [B*Ho*Wo, Cout] row-major IS the channels-last image
        return y.view(B, Ho, Wo, cout).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xl, offset, mask, weight, col = ctx.saved_tensors
        B, H, W, C, kh, kw = ctx.geom[:6]
        cout = weight.shape[0]
        Ho, Wo = ctx.out_hw
        L = _lib.lib()
        g = gy.permute(0, 2, 3, 1).reshape(B * Ho * Wo, cout)
        if not g.is_contiguous():
            g = g.contiguous()
        wmat = weight.permute(0, 2, 3, 1).reshape(cout, kh * kw * C)
        n_pix, K = B * Ho * Wo, kh * kw
        col_amax = ctx.col_amax
        if col is None:          # not kept (large maps): sample again
            col = torch.empty((n_pix, K * C), dtype=torch.float32, device=g.device)
            if ctx.matrix:
                from . import dense_conv
                col_amax = dense_conv.new_amax(g.device)
            check(L.gga_dcn_im2col_amax(F._p(xl), F._p(offset), F._p(mask), *ctx.geom, F._p(col), F._p(col_amax), F._stream()),
                  'gga_dcn_im2col')
        gw = gb = gx = None
        if ctx.matrix:
            from . import strided_conv
            fwd, _, _, bwd, bmask, bperm = _maps(n_pix, K, g.device)
            g_amax = strided_conv._amax(g)
            if ctx.needs_input_grad[3]:       # gw[k] = sum_p col[p * K + k]^T g[p]
                gw_kio = strided_conv._wgrad(col.view(n_pix * K, C), g, fwd, n_pix, col_amax, g_amax)
                gw = gw_kio.view(kh, kw, C, cout).permute(3, 2, 0, 1)
            del col
            # grad_col[p * K + k] = g[p] W_k^T: every row of grad_col runs the one tap of its position
            w_koi = weight.detach().permute(2, 3, 0, 1).reshape(K, cout, C).contiguous()
            gcol = strided_conv._apply(g, bwd, bmask, bperm, w_koi, n_pix * K, g_amax).view(n_pix, K * C)
        else:
            if ctx.needs_input_grad[3]:
                gw = (g.t() @ col).view(cout, kh, kw, C).permute(0, 3, 1, 2)
            del col
            gcol = g @ wmat
        if ctx.has_bias and ctx.needs_input_grad[4]:
            gb = F.channel_sums(g)
        goff, gmask = torch.empty_like(offset), torch.empty_like(mask)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(xl)          # channels-last like x
        check(L.gga_dcn_col2im(F._p(xl), F._p(offset), F._p(mask), F._p(gcol), *ctx.geom, F._p(gx), F._p(goff), F._p(gmask),
                               F._stream()), 'gga_dcn_col2im')
        return gx, goff, gmask, gw, gb, None, None, None


def ctx_keep_col(col, limit_bytes=2 << 30):
    """Keep the sampled columns for the weight gradient when they are small; above 2 GiB they are
    recomputed in backward (one more im2col pass instead of holding K*C floats per pixel)."""
    return col.numel() * 4 <= limit_bytes


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deform_groups=1):
    if groups != 1 or deform_groups != 1:
        raise NotImplementedError('DCNv2 with groups / deform_groups > 1 is not used by the GGA configs')
    return _ModulatedDeformConv.apply(x, offset, mask, weight, bias, _pair(stride), _pair(padding), _pair(dilation))


@CONV_LAYERS.register_module('DCNv2')
class ModulatedDeformConv2dPack(nn.Module):
    _version = 2

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deform_groups=1,
                 bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
        self.groups, self.deform_groups = groups, deform_groups
        self.transposed, self.output_padding = False, (0, 0)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.conv_offset = nn.Conv2d(in_channels, deform_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=self.stride, padding=self.padding,
                                     dilation=self.dilation, bias=True)
        self.init_weights()

    def init_weights(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()
        self.conv_offset.weight.data.zero_()         # starts as a plain convolution with mask 0.5
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        out = self.conv_offset(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        return modulated_deform_conv2d(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                       self.groups, self.deform_groups)
