"""Dense-convolution restatement of the sparse 3D convolutions of SparseEncoder, with autograd.
TEST INFRASTRUCTURE ONLY (see oracle/gga_oracle.c header).

The reference takes SubMConv3d / SparseConv3d from the un-vendored mmcv / spconv wheels and its
only test of them is shape-only and CUDA-gated (tests/test_models/test_common_modules/
test_middle_encoders.py:8-27), so sparse-conv numerics are PARITY-UNPINNED against the
reference. What is checked instead is the published definition of the two ops:

* SparseConv3d  = dense conv3d of the densified input; an output cell is active iff its
  receptive field contains an active input cell (conv3d of the occupancy mask > 0);
* SubMConv3d    = dense conv3d (stride 1, pad k//2) evaluated only at the active input cells.

``conv_ref`` densifies (small grids only: the dense tensor must fit on the CPU).
``conv_ref_pairs`` evaluates the SAME definition without a dense tensor - for every kernel
offset the (input row, output row) pairs are found by coordinate arithmetic + a sorted-key
lookup and ``y[out] += x[in] @ W[k]`` - so it runs at the reference's real grid
(41 x 1600 x 1408, 10^5 sites); tests/test_oracle.py checks the two against each other on
small grids. Neither imports the product: modules are dispatched on their attributes
(``subm``, ``kernel_size``, ``conv1`` ...), and the weight layout ``[kz,ky,kx,Cin,Cout]`` is
the one mmcv documents (mmdet3d/ops/spconv/overwrite_spconv/write_spconv2.py:47-48).
"""
import torch
import torch.nn.functional as TF


def _dense(feats, coors, batch, shape):
    D, H, W = shape
    x = feats.new_zeros(batch, feats.shape[1], D, H, W)
    c = coors.long()
    x[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = feats
    return x


def _w_dense(conv):   # [kz,ky,kx,Cin,Cout] -> [Cout,Cin,kz,ky,kx]
    return conv.weight.permute(4, 3, 0, 1, 2).contiguous()


def conv_ref(conv, feats, coors, batch, shape):
    """-> (features [N_out, Cout], coors [N_out, 4], out shape)."""
    x = _dense(feats, coors, batch, shape)
    if conv.subm:
        pad = tuple(k // 2 for k in conv.kernel_size)
        y = TF.conv3d(x, _w_dense(conv), padding=pad)
        c = coors.long()
        return y[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]], coors, shape
    occ = _dense(torch.ones(len(coors), 1), coors, batch, shape)
    ones = torch.ones(1, 1, *conv.kernel_size)
    act = TF.conv3d(occ, ones, stride=conv.stride, padding=conv.padding) > 0
    y = TF.conv3d(x, _w_dense(conv), stride=conv.stride, padding=conv.padding)
    oc = act[:, 0].nonzero()
    return y[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]], oc.int(), tuple(y.shape[2:])


def _key(c, shape):
    D, H, W = shape
    c = c.long()
    return ((c[:, 0] * D + c[:, 1]) * H + c[:, 2]) * W + c[:, 3]


def conv_ref_pairs(conv, feats, coors, batch, shape):
    """The same two definitions without a dense tensor. -> (features, coors, out shape); output
    rows in ascending (b, z, y, x) order for SparseConv3d, in input order for SubMConv3d."""
    k3, s3, p3 = tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding)
    c = coors.long()
    w = conv.weight.reshape(-1, conv.weight.shape[-2], conv.weight.shape[-1])
    offs = [(a, b, d) for a in range(k3[0]) for b in range(k3[1]) for d in range(k3[2])]
    if conv.subm:
        s3, p3 = (1, 1, 1), tuple(k // 2 for k in k3)
        oshape = tuple(shape)
        out_c = c
    else:
        oshape = tuple((shape[i] + 2 * p3[i] - k3[i]) // s3[i] + 1 for i in range(3))
        cand = []
        for (a, b, d) in offs:      # output o reads input o*s - p + k: every input proposes its outputs
            num = torch.stack([c[:, 1] + p3[0] - a, c[:, 2] + p3[1] - b, c[:, 3] + p3[2] - d], 1)
            ok = torch.ones(len(c), dtype=torch.bool, device=c.device)
            for i in range(3):
                ok &= (num[:, i] % s3[i] == 0) & (num[:, i] >= 0) & (num[:, i] // s3[i] < oshape[i])
            o = torch.stack([c[ok, 0]] + [num[ok, i] // s3[i] for i in range(3)], 1)
            cand.append(o)
        cand = torch.cat(cand)
        keys = torch.unique(_key(cand, oshape))          # sorted ascending
        D, H, W = oshape
        out_c = torch.stack([keys // (D * H * W), keys // (H * W) % D, keys // W % H, keys % W], 1)
    in_keys = _key(c, shape)
    order = torch.argsort(in_keys)
    sorted_keys = in_keys[order]
    y = feats.new_zeros(len(out_c), w.shape[-1])
    for ki, (a, b, d) in enumerate(offs):
        src = torch.stack([out_c[:, 0], out_c[:, 1] * s3[0] - p3[0] + a, out_c[:, 2] * s3[1] - p3[1] + b,
                           out_c[:, 3] * s3[2] - p3[2] + d], 1)
        ok = ((src[:, 1] >= 0) & (src[:, 1] < shape[0]) & (src[:, 2] >= 0) & (src[:, 2] < shape[1])
              & (src[:, 3] >= 0) & (src[:, 3] < shape[2]))
        sk = _key(src[ok], shape)
        pos = torch.searchsorted(sorted_keys, sk).clamp(max=len(sorted_keys) - 1)
        hit = sorted_keys[pos] == sk
        out_rows = ok.nonzero()[:, 0][hit]
        in_rows = order[pos[hit]]
        if len(out_rows):
            y = y.index_add(0, out_rows, feats[in_rows] @ w[ki])
    return y, out_c.int(), oshape


def _is_conv(m):
    return hasattr(m, 'subm') and hasattr(m, 'kernel_size') and hasattr(m, 'weight')


class Trace(list):
    """``trace`` argument that also records every convolution's INPUT features, coordinates and shape (``.inputs``)."""

    def __init__(self):
        super().__init__()
        self.inputs = []


def _run(module, feats, coors, batch, shape, conv_fn, trace):
    if _is_conv(module):
        if isinstance(trace, Trace):
            trace.inputs.append((feats, coors, shape))
        f, coors, shape = conv_fn(module, feats, coors, batch, shape)
        if trace is not None:
            trace.append((f, coors, shape))
        return f, coors, shape
    if hasattr(module, 'conv1') and hasattr(module, 'conv2') and hasattr(module, 'norm1'):      # SparseBasicBlock
        identity = feats
        f, coors, shape = _run(module.conv1, feats, coors, batch, shape, conv_fn, trace)
        f = torch.relu(module.norm1(f))
        f, coors, shape = _run(module.conv2, f, coors, batch, shape, conv_fn, trace)
        f = module.norm2(f)
        return torch.relu(f + identity), coors, shape
    if not isinstance(module, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(module, torch.nn.ReLU) \
            and len(module._modules):                                                                  # SparseSequential
        for m in module._modules.values():
            feats, coors, shape = _run(m, feats, coors, batch, shape, conv_fn, trace)
        return feats, coors, shape
    return module(feats), coors, shape       # BatchNorm1d / ReLU on the feature rows


def sparse_encoder_reference(encoder, voxel_features, coors, batch_size, pairs=False, trace=None):
    """SparseEncoder.forward (sparse_encoder.py:107-138) with the dense (default) or the
    pair-list restatement of every convolution. CPU tensors. ``trace``: list that receives
    (features, coors, shape) after every convolution, in execution order."""
    conv_fn = conv_ref_pairs if pairs else conv_ref
    shape = tuple(encoder.sparse_shape)
    f, c, shape = _run(encoder.conv_input, voxel_features, coors, batch_size, shape, conv_fn, trace)
    for layer in encoder.encoder_layers:
        f, c, shape = _run(layer, f, c, batch_size, shape, conv_fn, trace)
    f, c, shape = _run(encoder.conv_out, f, c, batch_size, shape, conv_fn, trace)
    x = _dense(f, c, batch_size, shape)
    N, C, D, H, W = x.shape
    return x.view(N, C * D, H, W), (f, c)
