"""Dense restatement of the MinkowskiEngine layers FCAF3D uses, with autograd. TEST INFRASTRUCTURE ONLY (see
oracle/gga_oracle.c header).

MinkowskiEngine is an un-vendored dependency of the reference (mmdet3d/models/backbones/mink_resnet.py:4-10 imports it
inside a try block; the reference's tests of these layers are CUDA-gated and shape-only), so their numerics are
PARITY-UNPINNED against the reference. What is checked is the published definition of each layer, evaluated here on a
DENSE grid with an occupancy mask - an independent formulation from the product's rule books (gga_amd/mink.py):

* a sparse tensor of tensor stride ts = (features [N, C], coordinates [N, 4] = batch + voxel coordinates, multiples of ts);
  ``dense()`` puts it on the lattice grid[b, :, x / ts, y / ts, z / ts] with occupancy 1;
* convolution k3 s1: conv3d(pad 1) read at the occupied cells; k3 s2: conv3d(stride 2, pad 1) read at the cells whose
  2 x 2 x 2 block holds an input (max_pool3d of the occupancy) - output o reads the inputs 2o - 1 .. 2o + 1; k1 s2:
  conv3d(kernel 1, stride 2);
* max pooling k2 s2: max_pool3d with unoccupied cells at -inf;
* generative transposed convolution k2 s2: conv_transpose3d(stride 2) - every input cell writes its 8 children;
* union addition: sum of the dense grids, occupancy OR;
* features_at_coordinates: trilinear interpolation on the lattice, absent cells as zeros.
Weight layout [kvol, Cin, Cout], offsets in (x, y, z) order with x slowest - the layout of gga_amd.mink.
"""
import torch
import torch.nn.functional as TF


def dense(feats, coords, ts, batch, shape):
    """-> (grid [B, C, X, Y, Z], occupancy [B, 1, X, Y, Z]); ``shape`` = lattice extent, coordinates must be >= 0."""
    g = feats.new_zeros((batch, feats.shape[1]) + tuple(shape))
    occ = feats.new_zeros((batch, 1) + tuple(shape))
    c = coords.long()
    i = (c[:, 0], slice(None), c[:, 1] // ts, c[:, 2] // ts, c[:, 3] // ts)
    g[i] = feats
    occ[c[:, 0], 0, c[:, 1] // ts, c[:, 2] // ts, c[:, 3] // ts] = 1
    return g, occ


def sparse(grid, occ, ts):
    """-> (features [N, C], coordinates [N, 4]) in ascending (b, x, y, z) order."""
    idx = (occ[:, 0] > 0).nonzero()
    return grid[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]], torch.cat([idx[:, :1], idx[:, 1:] * ts], 1)


def _w(weight, k):          # [k^3, Cin, Cout] -> [Cout, Cin, k, k, k]
    return weight.view(k, k, k, weight.shape[1], weight.shape[2]).permute(4, 3, 0, 1, 2)


def conv(grid, occ, weight, kernel, stride):
    if stride == 1:
        return TF.conv3d(grid, _w(weight, kernel), padding=kernel // 2) * occ, occ
    out_occ = TF.max_pool3d(occ, 2, 2, ceil_mode=True)
    y = TF.conv3d(grid, _w(weight, kernel), stride=2, padding=kernel // 2)
    y = y[:, :, :out_occ.shape[2], :out_occ.shape[3], :out_occ.shape[4]]
    if y.shape[2:] != out_occ.shape[2:]:      # conv3d floors the output extent where max_pool3d(ceil_mode) does not
        y = TF.pad(y, [0, out_occ.shape[4] - y.shape[4], 0, out_occ.shape[3] - y.shape[3], 0, out_occ.shape[2] - y.shape[2]])
    return y * out_occ, out_occ


def max_pool(grid, occ):
    g = torch.where(occ > 0, grid, torch.full_like(grid, float('-inf')))
    out_occ = TF.max_pool3d(occ, 2, 2, ceil_mode=True)
    y = TF.max_pool3d(g, 2, 2, ceil_mode=True)
    return torch.where(out_occ > 0, y, torch.zeros_like(y)), out_occ


def conv_transpose(grid, occ, weight):
    w = weight.view(2, 2, 2, weight.shape[1], weight.shape[2]).permute(3, 4, 0, 1, 2)       # [Cin, Cout, 2, 2, 2]
    y = TF.conv_transpose3d(grid * occ, w, stride=2)
    return y, TF.interpolate(occ, scale_factor=2, mode='nearest')


def union(g1, o1, g2, o2):
    return g1 * o1 + g2 * o2, torch.clamp(o1 + o2, max=1)


def features_at(grid, occ, ts, query):
    """``query`` [M, 4] float (batch, voxel coordinates) -> [M, C]."""
    b = query[:, 0].long()
    x = query[:, 1:] / ts
    lo = torch.floor(x)
    fr = x - lo
    out = grid.new_zeros((len(query), grid.shape[1]))
    ext = torch.tensor(grid.shape[2:], device=grid.device)
    for a in (0, 1):
        for c in (0, 1):
            for d in (0, 1):
                corner = torch.tensor([a, c, d], dtype=lo.dtype, device=grid.device)
                w = torch.prod(torch.where(corner.bool(), fr, 1 - fr), 1)
                p = (lo + corner).long()
                ok = ((p >= 0) & (p < ext)).all(1)
                p = torch.where(ok[:, None], p, torch.zeros_like(p))
                f = grid[b, :, p[:, 0], p[:, 1], p[:, 2]] * occ[b, 0, p[:, 0], p[:, 1], p[:, 2]][:, None]
                out = out + (w * ok)[:, None] * f
    return out


def instance_norm(feats, batch_index, n_batch, weight, bias, eps=1e-6):
    out = torch.empty_like(feats)
    for b in range(n_batch):
        m = batch_index == b
        f = feats[m]
        mean = f.mean(0, keepdim=True)
        var = ((f - mean) ** 2).mean(0, keepdim=True)
        out[m] = (f - mean) / torch.sqrt(var + eps)
    return out * weight + bias
