"""Generalised sparse tensors and layers with MinkowskiEngine's semantics - the subset the reference's FCAF3D trunk uses
(mmdet3d/models/backbones/mink_resnet.py:18-114, mmdet3d/models/dense_heads/fcaf3d_head.py:76-147,220-268,
mmdet3d/models/detectors/mink_single_stage.py:47-60): ``SparseTensor`` with a tensor stride, ``MinkowskiConvolution``
(kernel 3 or 1, stride 1 or 2), ``MinkowskiGenerativeConvolutionTranspose`` (kernel = stride = 2), ``MinkowskiMaxPooling``
(kernel = stride = 2), ``MinkowskiPruning``, ``MinkowskiBatchNorm`` / ``InstanceNorm`` / ``ReLU`` / ``ELU``, the union
addition of two tensors, ``features_at_coordinates`` and the per-sample decompositions.

MinkowskiEngine is an un-vendored dependency (no source under the reference tree, every test of it CUDA-gated and shape-only):
the semantics are restated from its published definition - **parity unpinned** - and checked against the dense restatement
oracle/mink_ref.py. With tensor stride ``ts`` (coordinates are multiples of it):

* convolution, stride 1, kernel 3: output coordinates = input coordinates, ``out[c] = sum_k in[c + (k - 1) ts] W[k]`` over
  the 27 offsets k in {0,1,2}^3 (first axis slowest); kernel 1: ``out[c] = in[c] W[0]``;
* convolution, stride 2: output coordinates = the distinct ``floor(c / 2ts) 2ts`` of the inputs (tensor stride 2 ts) and
  ``out[C] = sum_k in[C + (k - 1) ts] W[k]`` (kernel 3) or ``in[C] W[0]`` (kernel 1) - the offsets are those of the INPUT stride;
* max pooling, kernel = stride = 2: same output coordinates, maximum over the inputs ``C + {0,1}^3 ts``;
* generative transposed convolution, kernel = stride = 2: every input coordinate c generates the 8 outputs
  ``c + o ts/2``, o in {0,1}^3 (tensor stride ts / 2), ``out[c + o ts/2] = in[c] W[o]``;
* ``a + b`` on different coordinate sets: the union, absent entries counting as zero;
* ``features_at_coordinates``: multilinear interpolation between the (up to 8) lattice points of the tensor's stride around
  the query, absent lattice points counting as zero;
* quantisation of input points: ``floor``, one point per voxel (MinkowskiEngine's default mode keeps an arbitrary one;
  here: the first in input order).

**Kernel-offset order, and what it means for checkpoints.** The 27 (8) offsets of a convolution (generative transposed
convolution) are numbered here with the FIRST spatial axis slowest: k = 9 k1 + 3 k2 + k3 (children o = 4 o1 + 2 o2 + o3);
oracle/mink_ref.py uses the same numbering, so the tests cannot tell it from any other. MinkowskiEngine's own ``kernel_region``
iterator is un-vendored and its order is NOT pinned here (as far as its published source reads, it advances the first
spatial axis FASTEST). Training from scratch is indifferent to the numbering (the initialisation is symmetric under it); a
``state_dict`` trained with MinkowskiEngine would load without an error and compute with permuted kernels. Until one tensor
from a real MinkowskiEngine convolution with an asymmetric kernel pins the order, **checkpoints are not interchangeable with
the reference's FCAF3D**; ``reorder_kernel_offsets`` converts a kernel between the two numberings for whoever has such a file.

Convolutions run on the gather-GEMM kernels of the sparse 3D trunk (``strided_conv._apply`` / ``_wgrad``: forward,
backward-data and the deterministic weight gradient) through rule books built by ``gga_sparse_rulebook`` on the hash index
of ``sparse._Level``; coordinates are kept per level in units of the tensor stride and shifted to be non-negative."""
import math

import torch
from torch import nn

from . import _lib
from . import functional as F
from ._lib import check
from .sparse import _Level, _Rulebook, _i3

def reorder_kernel_offsets(kernel, kernel_size):
    """``kernel`` [kernel_size^3, Cin, Cout] numbered with the first spatial axis FASTEST (k = k1 + K k2 + K^2 k3) -> the
    numbering of this module (first axis slowest), or back: the permutation is its own inverse."""
    K = int(kernel_size)
    idx = torch.arange(K ** 3, device=kernel.device).reshape(K, K, K).permute(2, 1, 0).reshape(-1)
    return kernel[idx]


ALIGN = 64          # the shift that makes coordinates non-negative is a multiple of this (the coarsest tensor stride used)


class CoordMap:
    """One coordinate set: ``coords`` int32 [N, 4] (batch, c1, c2, c3) in units of ``stride`` (already shifted by
    ``origin`` voxels per axis, so they are >= 0), the hash index over them and the kernel maps cached per relation."""

    def __init__(self, coords, stride, batch_size, origin, extent):
        self.coords = coords.contiguous()
        self.stride, self.batch_size, self.origin = int(stride), int(batch_size), origin
        self.extent = tuple(int(e) for e in extent)            # exclusive upper bound of the level coordinates per axis
        self.n = int(coords.shape[0])
        self.level = _Level(self.coords, self.extent, batch_size) if self.n else None
        self.cache = {}

    def absolute(self):
        """[N, 4] int64: batch index and the coordinates in voxels of the input quantisation (as MinkowskiEngine's ``C``)."""
        c = self.coords.long()
        o = torch.as_tensor(self.origin, device=c.device)
        return torch.cat([c[:, :1], c[:, 1:] * self.stride - o], 1)

    def keys(self, coords=None):
        c = (self.coords if coords is None else coords).long()
        e = self.extent
        return ((c[:, 0] * e[0] + c[:, 1]) * e[1] + c[:, 2]) * e[2] + c[:, 3]

    # ---- derived maps ------------------------------------------------------------------------------------------------
    def parents(self):
        """Map of tensor stride 2 * stride holding the distinct ``coords // 2`` (ascending key order) + the parent row of
        every row of this map."""
        hit = self.cache.get('parents')
        if hit is None:
            pc = torch.cat([self.coords[:, :1], self.coords[:, 1:] // 2], 1)
            ext = tuple((e + 1) // 2 for e in self.extent)
            key = ((pc[:, 0].long() * ext[0] + pc[:, 1]) * ext[1] + pc[:, 2]) * ext[2] + pc[:, 3]
            uk, inv = torch.unique(key, return_inverse=True)
            out = torch.stack([uk // (ext[0] * ext[1] * ext[2]), uk // (ext[1] * ext[2]) % ext[0], uk // ext[2] % ext[1], uk % ext[2]], 1).int()
            hit = self.cache['parents'] = (CoordMap(out, self.stride * 2, self.batch_size, self.origin, ext), inv)
        return hit

    def children(self):
        """Map of tensor stride stride / 2 with the 8 generated coordinates of every row (row 8 i + o, o = 4 o1 + 2 o2 + o3)."""
        hit = self.cache.get('children')
        if hit is None:
            assert self.stride % 2 == 0, 'a generative transposed convolution halves the tensor stride'
            off = torch.tensor([[0, a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], dtype=torch.int32, device=self.coords.device)
            base = torch.cat([self.coords[:, :1], self.coords[:, 1:] * 2], 1)
            out = (base[:, None, :] + off[None]).reshape(-1, 4)
            hit = self.cache['children'] = CoordMap(out, self.stride // 2, self.batch_size, self.origin, tuple(2 * e for e in self.extent))
        return hit

    def subset(self, mask):
        return CoordMap(self.coords[mask], self.stride, self.batch_size, self.origin, self.extent)

    def lookup(self, coords):
        """Row of every coordinate of ``coords`` (level units, int) in this map, -1 where absent."""
        if self.n == 0:
            return torch.full((coords.shape[0],), -1, dtype=torch.long, device=coords.device)
        if 'sorted' not in self.cache:
            k = self.keys()
            order = torch.argsort(k)
            self.cache['sorted'] = (k[order], order)
        sk, order = self.cache['sorted']
        c = coords.long()
        ok = (c[:, 1:] >= 0).all(1) & (c[:, 1] < self.extent[0]) & (c[:, 2] < self.extent[1]) & (c[:, 3] < self.extent[2])
        q = self.keys(torch.where(ok[:, None], c, torch.zeros_like(c)))
        pos = torch.searchsorted(sk, q).clamp(max=len(sk) - 1)
        hit = ok & (sk[pos] == q)
        return torch.where(hit, order[pos], torch.full_like(pos, -1))

    # ---- kernel maps: (forward rule book [kvol, n_out], backward rule book [kvol, n_in]) -------------------------------
    def kernel_map(self, out_map, kernel, stride):
        key = ('kmap', id(out_map), kernel, stride)
        hit = self.cache.get(key)
        if hit is None:
            if self.n == 0 or out_map.n == 0:
                hit = (None, None)
            elif kernel == 3 and stride == 1:
                rb = self.level.subm_rulebook((3, 3, 3))
                hit = (rb, _Rulebook(rb.nbr.flip(0).contiguous()))       # SubM: the transposed map is the reversed offsets
            else:
                L = _lib.lib()
                kvol = kernel ** 3
                pad = kernel // 2
                self.level.ensure_index(), out_map.level.ensure_index()
                nbr = torch.empty((kvol, out_map.n), dtype=torch.int32, device=self.coords.device)
                nbr_t = torch.empty((kvol, self.n), dtype=torch.int32, device=self.coords.device)
                check(L.gga_sparse_rulebook(F._p(out_map.coords), out_map.n, F._p(self.coords), self.n, self.batch_size,
                                            _i3(self.extent), _i3(out_map.extent), _i3((kernel,) * 3), _i3((stride,) * 3), _i3((pad,) * 3),
                                            F._p(self.level.index), self.level.index_n, F._p(out_map.level.index), out_map.level.index_n,
                                            F._p(nbr), F._p(nbr_t), F._stream()), 'gga_sparse_rulebook')
                hit = (_Rulebook(nbr), _Rulebook(nbr_t))
            self.cache[key] = hit
        return hit

    def transpose_map(self, child_map):
        """Generative transposed convolution: child row 8 i + o reads row i through offset o."""
        key = ('tmap', id(child_map))
        hit = self.cache.get(key)
        if hit is None:
            dev = self.coords.device
            rows = torch.arange(self.n, device=dev, dtype=torch.int32)
            nbr = torch.full((8, self.n, 8), -1, dtype=torch.int32, device=dev)
            nbr_t = torch.empty((8, self.n), dtype=torch.int32, device=dev)
            for o in range(8):
                nbr[o, :, o] = rows
                nbr_t[o] = rows * 8 + o
            hit = self.cache[key] = (_Rulebook(nbr.reshape(8, self.n * 8).contiguous()), _Rulebook(nbr_t))
        return hit


class SparseTensor:
    """``features`` [N, C] on the coordinates of ``cmap`` (``C`` = [N, 4]: batch index, integer voxel coordinates)."""

    def __init__(self, features=None, coordinates=None, cmap=None, batch_size=None):
        if cmap is None:
            cmap, features = quantize(coordinates, features, batch_size)
        self.F, self.cmap = features, cmap

    features = property(lambda self: self.F)
    tensor_stride = property(lambda self: self.cmap.stride)

    @property
    def C(self):
        return self.cmap.absolute().int()

    def replace(self, features, cmap=None):
        return SparseTensor(features, cmap=cmap or self.cmap)

    @property
    def decomposition_permutations(self):
        hit = self.cmap.cache.get('perms')
        if hit is None:
            b = self.cmap.coords[:, 0]
            hit = self.cmap.cache['perms'] = [torch.nonzero(b == i).squeeze(1) for i in range(self.cmap.batch_size)]
        return hit

    @property
    def decomposed_coordinates(self):
        c = self.cmap.absolute()
        return [c[p, 1:] for p in self.decomposition_permutations]

    def features_at_coordinates(self, query):
        """``query`` [M, 4] float (batch index, voxel coordinates) -> [M, C]: multilinear interpolation on this tensor's lattice."""
        m = self.cmap
        ts = float(m.stride)
        o = torch.as_tensor(m.origin, device=query.device, dtype=query.dtype)
        x = (query[:, 1:] + o) / ts                              # level units, fractional
        lo = torch.floor(x)
        frac = x - lo
        b = query[:, :1].long()
        out = self.F.new_zeros((query.shape[0], self.F.shape[1]))
        for a in (0, 1):
            for bb in (0, 1):
                for c in (0, 1):
                    corner = torch.tensor([a, bb, c], device=query.device, dtype=lo.dtype)
                    w = torch.prod(torch.where(corner.bool(), frac, 1 - frac), dim=1)
                    row = m.lookup(torch.cat([b, (lo + corner).long()], 1))
                    ok = (row >= 0) & (w > 0)
                    out = out + torch.where(ok, w, torch.zeros_like(w))[:, None] * self.F[row.clamp(min=0)]
        return out

    def __add__(self, other):
        if other.cmap is self.cmap:
            return self.replace(self.F + other.F)
        a, b = self.cmap, other.cmap
        assert a.stride == b.stride and a.extent == b.extent and a.origin == b.origin, 'union of tensors of one lattice only'
        keys = torch.cat([a.keys(), b.keys()])
        uk, inv = torch.unique(keys, return_inverse=True)
        e = a.extent
        coords = torch.stack([uk // (e[0] * e[1] * e[2]), uk // (e[1] * e[2]) % e[0], uk // e[2] % e[1], uk % e[2]], 1).int()
        feats = self.F.new_zeros((len(uk), self.F.shape[1])).index_add(0, inv[:a.n], self.F).index_add(0, inv[a.n:], other.F)
        return SparseTensor(feats, cmap=CoordMap(coords, a.stride, a.batch_size, a.origin, a.extent))


def quantize(coordinates, features, batch_size=None):
    """``coordinates`` [N, 4] (batch index, float or int voxel coordinates) -> (CoordMap of tensor stride 1, features of the kept
    points): floor, then the FIRST point of every voxel in input order."""
    c = torch.floor(coordinates.float()).long() if coordinates.is_floating_point() else coordinates.long()
    B = int(batch_size) if batch_size is not None else (int(c[:, 0].max()) + 1 if len(c) else 1)
    if len(c) == 0:
        return CoordMap(c.int(), 1, B, (0, 0, 0), (ALIGN, ALIGN, ALIGN)), features
    lo, hi = c[:, 1:].min(0).values, c[:, 1:].max(0).values
    origin = tuple(int(-(-max(0, -int(v)) // ALIGN) * ALIGN) for v in lo)         # multiples of ALIGN: floor(c / s) keeps its meaning
    shifted = c[:, 1:] + torch.tensor(origin, device=c.device)
    extent = tuple(int(-(-(int(h) + o + 1) // ALIGN) * ALIGN) for h, o in zip(hi, origin))
    key = ((c[:, 0] * extent[0] + shifted[:, 0]) * extent[1] + shifted[:, 1]) * extent[2] + shifted[:, 2]
    uk, inv = torch.unique(key, return_inverse=True)
    first = torch.full((len(uk),), len(c), dtype=torch.long, device=c.device).scatter_reduce(0, inv, torch.arange(len(c), device=c.device), 'amin')
    first = torch.sort(first).values                                             # kept points in input order
    coords = torch.cat([c[first, :1], shifted[first]], 1).int()
    return CoordMap(coords, 1, B, origin, extent), features[first]


def batch_sparse_collate(data, device=None):
    """``ME.utils.batch_sparse_collate``: list of (coordinates [n, 3], features [n, C]) -> ([N, 4] float with the batch index
    in front, [N, C])."""
    coords = torch.cat([torch.cat([c.new_full((len(c), 1), i), c], 1) for i, (c, _) in enumerate(data)])
    feats = torch.cat([f for _, f in data])
    return (coords.to(device), feats.to(device)) if device is not None else (coords, feats)


# ----------------------------------------------------------------------------------------------------------- convolution
class _MapConv(torch.autograd.Function):
    """y [n_out, cout] = sum_k x[fwd[k]] W[k] through the rule books (fwd [kvol, n_out], bwd [kvol, n_in])."""

    @staticmethod
    def forward(ctx, x, weight, fwd, bwd, n_out, flip_bwd):
        from . import strided_conv as S
        x = x.contiguous()
        w = weight.detach().contiguous()
        # the layer's own parameter (not the zero-padded temporary of odd widths, map_conv): its packed operands live in the
        # weight bank and are refreshed with all others once per optimizer step
        banked = weight.grad_fn is None and w.data_ptr() == weight.data_ptr()
        y = S._apply(x, fwd.nbr, fwd.mask, fwd.perm, w, n_out, bank=banked)
        ctx.save_for_backward(x, weight)
        ctx.maps, ctx.flip, ctx.banked = (fwd, bwd, n_out), flip_bwd, banked
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import strided_conv as S
        x, weight = ctx.saved_tensors
        fwd, bwd, n_out = ctx.maps
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            # gx[i] = sum_k gy[bwd[k][i]] W[k]^T: bwd[k][i] is the output row that reads row i through offset k (for SubM maps
            # that is row K-1-k of the forward rule book, which is how CoordMap.kernel_map builds it)
            wt = weight.detach().transpose(1, 2)                # [kvol, cout, cin]: a view for the bank, a copy otherwise
            gx = S._apply(gy, bwd.nbr, bwd.mask, bwd.perm, wt if ctx.banked else wt.contiguous(), x.shape[0], bank=ctx.banked)
        if ctx.needs_input_grad[1]:
            gw = S._wgrad(x, gy, fwd.nbr, n_out)
        return gx, gw, None, None, None, None


def _plain_conv(x, weight, fwd):
    """The same product in plain torch for widths the gather kernels do not take (C % 4 != 0, e.g. 3 input channels or a
    1-channel output) - differentiable through index_add / matmul."""
    kvol, n_out = fwd.nbr.shape
    y = x.new_zeros((n_out, weight.shape[2]))
    for k in range(kvol):
        idx = fwd.nbr[k].long()
        ok = (idx >= 0).nonzero().squeeze(1)
        if len(ok):
            y = y.index_add(0, ok, x[idx[ok]] @ weight[k])
    return y


def map_conv(x, weight, fwd, bwd, n_out, subm):
    cin, cout = weight.shape[1], weight.shape[2]
    if not x.is_cuda:
        return _plain_conv(x, weight, fwd)
    pi, po = -cin % 4, -cout % 4
    if pi or po:
        # widths the gather kernels do not take as they are (3 input channels; 1, 6 or 10 output channels of the head): zero
        # columns up to the next multiple of 4 - the extra products are zeros, the extra outputs are dropped, and autograd
        # slices the gradients back. (The per-offset index / matmul / index_add loop of _plain_conv cost 83 launches of
        # torch's indexing backward per FCAF3D step: 49 of the step's 214 ms of kernels.)
        xp = torch.nn.functional.pad(x, (0, pi)) if pi else x
        wp = torch.nn.functional.pad(weight, (0, po, 0, pi))
        return _MapConv.apply(xp, wp, fwd, bwd, n_out, subm)[:, :cout]
    return _MapConv.apply(x, weight, fwd, bwd, n_out, subm)


class MinkowskiConvolution(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and dilation == 1 and kernel_size in (1, 3) and stride in (1, 2)
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_channels, out_channels, kernel_size, stride
        self.kernel = nn.Parameter(torch.empty(kernel_size ** 3, in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None
        # MinkowskiEngine's reset_parameters: uniform(-stdv, stdv), stdv = 1 / sqrt(in_channels * kernel_volume)
        stdv = 1.0 / math.sqrt(in_channels * kernel_size ** 3)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if bias:
                self.bias.uniform_(-stdv, stdv)

    def forward(self, x):
        m = x.cmap
        out_map = m if self.stride == 1 else m.parents()[0]
        if m.n == 0:
            y = x.F.new_zeros((0, self.out_channels)) + 0 * self.kernel.sum()
        elif self.kernel_size == 1 and self.stride == 1:
            y = x.F @ self.kernel[0]
        else:
            fwd, bwd = m.kernel_map(out_map, self.kernel_size, self.stride)
            y = map_conv(x.F, self.kernel, fwd, bwd, out_map.n, self.kernel_size == 3 and self.stride == 1)
        if self.bias is not None:
            y = y + self.bias
        return SparseTensor(y, cmap=out_map)


class MinkowskiGenerativeConvolutionTranspose(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and kernel_size == 2 and stride == 2
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel = nn.Parameter(torch.empty(8, in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None
        stdv = 1.0 / math.sqrt(in_channels * 8)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    def forward(self, x):
        m = x.cmap
        child = m.children()
        if m.n == 0:
            y = x.F.new_zeros((0, self.out_channels)) + 0 * self.kernel.sum()
        else:
            fwd, bwd = m.transpose_map(child)
            y = map_conv(x.F, self.kernel, fwd, bwd, child.n, False)
        if self.bias is not None:
            y = y + self.bias
        return SparseTensor(y, cmap=child)


class MinkowskiMaxPooling(nn.Module):
    def __init__(self, kernel_size=2, stride=2, dimension=3):
        super().__init__()
        assert dimension == 3 and kernel_size == 2 and stride == 2

    def forward(self, x):
        out_map, parent = x.cmap.parents()
        idx = parent[:, None].expand(-1, x.F.shape[1])
        y = x.F.new_full((out_map.n, x.F.shape[1]), float('-inf')).scatter_reduce(0, idx, x.F, 'amax', include_self=True)
        return SparseTensor(y, cmap=out_map)


class _TakeRows(torch.autograd.Function):
    """t[idx] for an index vector WITHOUT duplicates: the backward copies the gradient rows into a zero tensor (torch's
    backward of advanced indexing sorts the indices and accumulates - 4.5 ms for a [930 k, 64] tensor, three times per FCAF3D
    step)."""

    @staticmethod
    def forward(ctx, t, idx):
        ctx.idx, ctx.shape = idx, t.shape
        return t.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        return g.new_zeros(ctx.shape).index_copy_(0, ctx.idx, g), None


def take_rows(t, idx):
    return _TakeRows.apply(t, idx) if t.requires_grad else t.index_select(0, idx)


class MinkowskiPruning(nn.Module):
    def forward(self, x, mask):
        keep = torch.nonzero(mask).squeeze(1)
        return SparseTensor(take_rows(x.F, keep), cmap=x.cmap.subset(mask))


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)

    def forward(self, x):
        return self.forward_act(x, relu=False)

    def forward_act(self, x, relu=False, residual=None):
        """``relu(bn(x) + residual)`` (each part optional) on the feature rows: the fused BatchNorm pass pair of the BEV trunk
        (``functional.bn_act``: statistics + apply with the ReLU, the residual and the consumer's absmax in one sweep; the
        eager ops where it does not apply - CPU tensors, odd widths)."""
        from . import functional as F
        if x.F.shape[0] == 0:
            return x
        res = None if residual is None else residual.F
        return x.replace(F.bn_act(x.F, self.bn, relu=relu, residual=res))


class MinkowskiInstanceNorm(nn.Module):
    """Per sample and channel: (x - mean) / sqrt(var + 1e-6) over the sample's points (biased variance), then weight, bias."""

    def __init__(self, num_features):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))
        self.eps = 1e-6

    def forward(self, x):
        # the per-sample sums and their broadcast back as products with the samples' indicator matrix [B, n]: plain GEMMs in
        # both directions (stats[b] with one index per point is, in torch's backward, a sort of all n indices per use: 4.5 ms
        # each at 930 k points - 48 of the FCAF3D step's 200 ms of kernels; index_add's is 64 n float atomics on B x C addresses)
        b = x.cmap.coords[:, 0].long()
        B = x.cmap.batch_size
        ind = x.F.new_zeros((B, x.F.shape[0])).scatter_(0, b[None, :], 1.0)
        cnt = ind.sum(1, keepdim=True).clamp(min=1)
        mean = (ind @ x.F) / cnt
        cen = x.F - ind.t() @ mean
        var = (ind @ (cen * cen)) / cnt
        return x.replace(cen * (ind.t() @ torch.rsqrt(var + self.eps)) * self.weight + self.bias)


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return x.replace(torch.relu(x.F))


class MinkowskiELU(nn.Module):
    def forward(self, x):
        return x.replace(nn.functional.elu(x.F))


class BasicBlock(nn.Module):
    """``MinkowskiEngine.modules.resnet_block.BasicBlock``: conv3(stride) - BN - ReLU - conv3 - BN, + (downsampled) input, ReLU
    (BatchNorm momentum 0.1)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=3):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.norm1.forward_act(self.conv1(x), relu=True)
        out = self.conv2(out)
        residual = x if self.downsample is None else self.downsample(x)
        if out.cmap is residual.cmap:                        # (always, in a residual block: same coordinate set, same row order)
            return self.norm2.forward_act(out, relu=True, residual=residual)
        return self.relu(self.norm2(out) + residual)
