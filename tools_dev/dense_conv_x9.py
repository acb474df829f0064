"""Dense 3x3 / stride-1 / pad-1 convolution on channels-last activations through the sparse conv
kernels: a dense image is the sparse tensor with every pixel active, so the bf16x9 gather-GEMM of
``csrc/sparse_conv.hip`` (fp32 products as nine exact bf16 partial products) applies with a rule
book that is pure index arithmetic. Used for the 64-channel 3x3 convolutions of the BEV trunk and
the task heads (``second.py:46-63``, ``centerpoint_head.py:60-90`` of the reference), where
MIOpen's fp32 implicit GEMM runs at ~100 TFLOP/s; the weight gradient stays on MIOpen.

EXPERIMENT (tools_dev/bench_dense_conv.py), not wired into the model: it ties MIOpen (107 vs 104
TFLOP/s at C=64, 113 vs 111 at C=128, results equal to 2e-6), so the product keeps MIOpen."""
import torch

from gga_amd import functional as F
from gga_amd import sparse as S

_MAPS = {}


class _DenseRulebook:
    """nbr [9, B*H*W] i32: row of the pixel at (y + ky - 1, x + kx - 1), -1 outside the image."""

    def __init__(self, B, H, W, device):
        rows = torch.arange(B * H * W, dtype=torch.int32, device=device).view(B, H, W)
        pad = torch.nn.functional.pad(rows, (1, 1, 1, 1), value=-1)
        self.nbr = torch.stack([pad[:, ky:ky + H, kx:kx + W].reshape(-1) for ky in range(3) for kx in range(3)]).contiguous()
        self.perm = self.mask = None            # every row uses every offset: nothing to sort or skip


def _rulebook(B, H, W, device):
    key = (B, H, W, str(device))
    rb = _MAPS.get(key)
    if rb is None:
        rb = _MAPS[key] = _DenseRulebook(B, H, W, device)
    return rb


class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        F._need_cuda(x, weight)
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        xr = x.permute(0, 2, 3, 1)
        assert xr.is_contiguous(), 'channels-last activations expected'
        xr = xr.reshape(-1, cin)
        w = weight.detach().permute(2, 3, 1, 0).reshape(9, cin, cout).contiguous()     # [k][ci][co]
        rb = _rulebook(B, H, W, x.device)
        y = torch.empty((B * H * W, cout), dtype=torch.float32, device=x.device)
        S._conv_apply(xr, rb, S._pack_weight(w, 9, cin, cout, 0, split=True), B * H * W, 9, cin, cout, 0, y)
        ctx.save_for_backward(x, weight)
        return y.view(B, H, W, cout).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            rb = _rulebook(B, H, W, x.device)
            w = weight.detach().permute(2, 3, 1, 0).reshape(9, cin, cout).contiguous()
            g = torch.empty((B * H * W, cin), dtype=torch.float32, device=x.device)
            # transposed map of a stride-1 'same' convolution = reversed offsets (as for SubM)
            S._conv_apply(gy.permute(0, 2, 3, 1).reshape(-1, cout), rb, S._pack_weight(w, 9, cout, cin, 1, split=True),
                          B * H * W, 9, cout, cin, 1, g)
            gx = g.view(B, H, W, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [False, True, False])[1]
        return gx, gw


def eligible(conv, x):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and conv.in_channels % 4 == 0 and conv.out_channels <= 128 and x.is_cuda and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last))


def conv3x3(x, weight):
    return _Conv3x3Fn.apply(x, weight)
