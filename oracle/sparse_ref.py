"""Dense-convolution restatement of the sparse 3D convolutions of SparseEncoder, with autograd.
TEST INFRASTRUCTURE ONLY (see oracle/gga_oracle.c header).

The reference takes SubMConv3d / SparseConv3d from the un-vendored mmcv / spconv wheels and its
only test of them is shape-only and CUDA-gated (tests/test_models/test_common_modules/
test_middle_encoders.py:8-27), so sparse-conv numerics are PARITY-UNPINNED against the
reference. What is checked instead is the published definition of the two ops:

* SparseConv3d  = dense conv3d of the densified input; an output cell is active iff its
  receptive field contains an active input cell (conv3d of the occupancy mask > 0);
* SubMConv3d    = dense conv3d (stride 1, pad k//2) evaluated only at the active input cells.

Small grids only (the dense tensor must fit comfortably on the CPU).
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


def _run(module, feats, coors, batch, shape):
    from gga_amd.sparse import SparseConvolution, SparseSequential
    from gga_amd.sparse_encoder import SparseBasicBlock
    if isinstance(module, SparseConvolution):
        return conv_ref(module, feats, coors, batch, shape)
    if isinstance(module, SparseBasicBlock):
        identity = feats
        f, coors, shape = conv_ref(module.conv1, feats, coors, batch, shape)
        f = torch.relu(module.norm1(f))
        f, coors, shape = conv_ref(module.conv2, f, coors, batch, shape)
        f = module.norm2(f)
        return torch.relu(f + identity), coors, shape
    if isinstance(module, SparseSequential):
        for m in module._modules.values():
            feats, coors, shape = _run(m, feats, coors, batch, shape)
        return feats, coors, shape
    return module(feats), coors, shape       # BatchNorm1d / ReLU on the feature rows


def sparse_encoder_reference(encoder, voxel_features, coors, batch_size):
    """SparseEncoder.forward (sparse_encoder.py:107-138) with dense convolutions. CPU tensors."""
    shape = tuple(encoder.sparse_shape)
    f, c, shape = _run(encoder.conv_input, voxel_features, coors, batch_size, shape)
    for layer in encoder.encoder_layers:
        f, c, shape = _run(layer, f, c, batch_size, shape)
    f, c, shape = _run(encoder.conv_out, f, c, batch_size, shape)
    x = _dense(f, c, batch_size, shape)
    N, C, D, H, W = x.shape
    return x.view(N, C * D, H, W), (f, c)
