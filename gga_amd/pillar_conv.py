"""First SECOND convolution on the pillar canvas, with its backward restricted to the pillars.

The canvas of ``PointPillarsScatter`` (pillar_scatter.py:58-102) is zero except at the occupied
cells (~7 % at KITTI sizes), and its gradient is read back only there (the scatter's backward is a
gather). The forward of the convolution that consumes it (second.py:49-57, block 0, conv 0) is the
plain dense convolution (run as strided_conv.py's gather-GEMM over the canvas rows); its backward needs

* the input gradient only at the occupied cells,
* the weight gradient ``sum_cells x^T gy``, where ``x`` is non-zero only at the occupied cells,

i.e. a gather-GEMM over the pillars with the gather map ``gga_pillar_conv_map`` — the same
kernels as the sparse 3D convolution (``gga_sparse_conv_apply_split`` / ``gga_sparse_conv_wgrad``),
with the output gradient (NHWC rows = output cells) as the gathered operand. At batch 16 that
replaces a 3.1 ms backward-data convolution into an 877 MB canvas gradient (+ its zero fill and
the scatter's gather) and a 3.3 ms weight-gradient convolution by < 0.5 ms of work; the values
are those of the dense backward up to fp32 summation order. The canvas itself is not kept for the
backward pass.
"""
import torch
from torch import nn

from . import _lib
from . import functional as F
from ._lib import check


class PillarSupport:
    """What a canvas was scattered from: pillar features [M,C], coors [M,4] (b,z,y,x) int32,
    optional device count of valid rows. Attached to the canvas by ``functional.pillar_scatter``
    when the cells are unique and the canvas is channels-last."""
    __slots__ = ('feats', 'coors', 'num_valid')

    def __init__(self, feats, coors, num_valid):
        self.feats, self.coors, self.num_valid = feats, coors, num_valid


def eligible(conv, canvas):
    sup = getattr(canvas, 'pillar_support', None)
    return (sup is not None and type(conv) is nn.Conv2d and conv.bias is None and conv.groups == 1
            and tuple(conv.dilation) == (1, 1) and conv.padding_mode == 'zeros' and not isinstance(conv.padding, str)
            and canvas.is_cuda and canvas.dtype == torch.float32 and torch.is_grad_enabled()
            and (sup.feats.requires_grad or conv.weight.requires_grad)
            and conv.in_channels <= 128 and conv.out_channels <= 128 and conv.in_channels % 4 == 0
            and canvas.is_contiguous(memory_format=torch.channels_last))


class _PillarConv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, weight, canvas, coors, num_valid, stride, padding):
        from . import strided_conv
        kh, kw = weight.shape[2:]
        if (strided_conv.ENABLED and kh == kw and stride[0] == stride[1] and padding[0] == padding[1] and stride[0] > 1
                and kh <= 5 and weight.shape[0] % 4 == 0):
            # the dense forward as a gather-GEMM over the canvas rows (strided_conv.py): no framework convolution
            B, Ci, ny, nx = canvas.shape
            bk = strided_conv.book(B, ny, nx, kh, stride[0], padding[0], canvas.device)
            # the canvas holds the pillar features and zeros: its largest magnitude is theirs (16 MB instead of 877 MB)
            y, stats = strided_conv._apply(strided_conv._rows(canvas), bk.fwd, bk.fwd_mask, bk.fwd_perm,
                                           weight.detach().permute(2, 3, 1, 0).reshape(kh * kw, Ci, -1), bk.n_out,
                                           strided_conv._amax(feats.detach().contiguous()), want_stats=True)
            y = y.view(B, bk.Ho, bk.Wo, -1).permute(0, 3, 1, 2)
        else:
            y = torch.conv2d(canvas, weight, None, stride, padding)
            stats = torch.empty(0, dtype=torch.float64, device=y.device)
        ctx.save_for_backward(feats, weight, coors, num_valid)
        ctx.geom = (tuple(canvas.shape), stride, padding)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return y, stats

    @staticmethod
    def backward(ctx, gy, _gstats=None):
        from .sparse import _Rulebook, _conv_apply, _pack_weight, conv_wgrad
        feats, w, coors, num_valid = ctx.saved_tensors
        (B, Ci, ny, nx), (sh, sw), (ph, pw) = ctx.geom
        Co, _, kh, kw = w.shape
        kvol, m = kh * kw, feats.shape[0]
        gy = gy.contiguous(memory_format=torch.channels_last)
        rows = gy.permute(0, 2, 3, 1).reshape(-1, Co)            # NHWC memory: a view, one row per output cell
        feats = feats.contiguous()
        L = _lib.lib()
        nbr = torch.empty((kvol, m), dtype=torch.int32, device=gy.device)
        check(L.gga_pillar_conv_map(F._p(coors), m, F._p(num_valid), B, ny, nx, kh, kw, sh, sw, ph, pw, F._p(nbr),
                                    F._stream()), 'gga_pillar_conv_map')
        gx = gw = None
        if ctx.needs_input_grad[0]:
            rb = _Rulebook(nbr)             # rows sorted by tap pattern (the parity class of the cell)
            wk = w.permute(2, 3, 0, 1).reshape(kvol, Co, Ci).contiguous()       # tap-major [k][co][ci]
            gx = torch.empty_like(feats)
            _conv_apply(rows, rb, _pack_weight(wk, kvol, Co, Ci, 0), m, kvol, Co, Ci, 0, gx)
        if ctx.needs_input_grad[1]:
            g = torch.empty((kvol, Co, Ci), dtype=torch.float32, device=gy.device)
            conv_wgrad(rows, feats, nbr, m, kvol, Co, Ci, g)
            gw = g.view(kh, kw, Co, Ci).permute(2, 3, 0, 1)
        return gx, gw, None, None, None, None, None


def pillar_conv2d(canvas, conv):
    """``conv(canvas)`` for a canvas carrying its ``pillar_support``; gradients flow to the pillar
    features directly (the scatter's own backward is bypassed) and to ``conv.weight``."""
    sup = canvas.pillar_support
    coors = sup.coors if sup.coors.dtype == torch.int32 else sup.coors.int()
    y, stats = _PillarConv2d.apply(sup.feats, conv.weight, canvas.detach(), coors.contiguous(), sup.num_valid,
                                   tuple(conv.stride), tuple(conv.padding))
    if stats.numel():
        F.attach_bn_partials(y, stats)       # per-channel sums of y for the BatchNorm that follows (functional.bn_act)
    return y
