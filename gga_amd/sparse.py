"""Sparse tensors and sparse 3D convolution layers on the HIP kernels of
gga_amd/csrc/sparse_conv.hip — the subset of the mmcv / spconv API the reference's
SparseEncoder uses (mmdet3d/ops/sparse_block.py:9-20,167-184; mmdet3d/models/
middle_encoders/sparse_encoder.py:120-138): ``SparseConvTensor(features, indices,
spatial_shape, batch_size)`` with ``.features`` / ``.indices`` / ``.dense()`` /
``replace_feature``, ``SparseModule``, ``SparseSequential`` and the conv layers
``SubMConv3d`` / ``SparseConv3d`` registered in ``CONV_LAYERS`` so
``build_conv_layer(dict(type='SubMConv3d', indice_key=...), Cin, Cout, k, stride=, padding=,
bias=False)`` works unchanged. Weights use the mmcv layout ``[kz, ky, kx, Cin, Cout]``.
"""
import ctypes as C
import os
import math

import torch
from torch import nn

from . import _lib
from . import functional as F
from ._lib import check
from .registry import CONV_LAYERS


def _triple(v):
    return tuple(int(x) for x in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


def _i3(v):
    return (C.c_int32 * 3)(*v)


def mask_order(mask, kvol):
    """Rows in ascending order of their offset mask, ties in row order: what ``torch.sort(mask, stable=True)[1].int()`` returns,
    from an LSD radix sort over the ``kvol`` mask bits (gga_sparse_mask_order, include/gga_hip.h) - the framework's stable
    sort of int32 keys is a 22-launch merge sort per rule book on this stack."""
    L = _lib.lib()
    n = int(mask.shape[0])
    order = torch.empty(n, dtype=torch.int32, device=mask.device)
    nbytes = int(L.gga_sparse_mask_order_workspace_bytes(n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=mask.device)
    check(L.gga_sparse_mask_order(F._p(mask), n, int(kvol), F._p(order), F._p(ws), nbytes, F._stream()), 'gga_sparse_mask_order')
    return order


class _Rulebook:
    """Gather map [kvol, n_rows] + per-row offset bit mask + mask-sorted processing order."""

    def __init__(self, nbr, coors=None, occupancy=0.0, extent=None):
        self.nbr = nbr
        kvol, n = nbr.shape
        self.mask = self.perm = None
        # submanifold rule books: the level's coordinates and the share of its grid cells that are active (halo form)
        self.coors, self.occupancy, self._halo = coors, occupancy, None
        self.extent = extent                    # (batch size, largest grid extent) of the level, for the Z-order tiling
        if kvol <= 32:
            self.mask = torch.empty(n, dtype=torch.int32, device=nbr.device)
            check(_lib.lib().gga_sparse_rowmask(F._p(nbr), n, kvol, F._p(self.mask), F._stream()), 'gga_sparse_rowmask')
            # index preprocessing (once per level, shared by every conv on it): rows with the same
            # neighbour pattern become adjacent, so a 128-row tile skips the offsets none of them uses
            self.perm = mask_order(self.mask, kvol)

    def halo(self):
        if self._halo is None:
            self._halo = _Halo(self.coors, self)
        return self._halo


def _spread3(v):
    """Bits of v (< 2^16, int64) moved to every third position."""
    v = v & 0xFFFF
    v = (v | (v << 32)) & 0x1F00000000FFFF
    v = (v | (v << 16)) & 0x1F0000FF0000FF
    v = (v | (v << 8)) & 0x100F00F00F00F00F
    v = (v | (v << 4)) & 0x10C30C30C30C30C3
    v = (v | (v << 2)) & 0x1249249249249249
    return v


def morton_order(coors, batch_size=None, max_extent=None):
    """Rows of ``coors`` [n,4] (batch, z, y, x) along a Z-order curve per sample: consecutive rows are close in space.
    With the level's ``batch_size`` and largest grid extent on a device tensor: gga_sparse_morton_order (one key kernel + a radix
    sort over the bits the extent can set; int32 order) - the expression below is 17 elementwise launches and a merge sort."""
    if batch_size is not None and max_extent is not None and coors.is_cuda and coors.dtype == torch.int32 and coors.shape[0]:
        L = _lib.lib()
        c = coors.contiguous()
        n = int(c.shape[0])
        order = torch.empty(n, dtype=torch.int32, device=c.device)
        nbytes = int(L.gga_sparse_morton_order_workspace_bytes(n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=c.device)
        check(L.gga_sparse_morton_order(F._p(c), n, int(batch_size), int(max_extent), F._p(order), F._p(ws), nbytes, F._stream()),
              'gga_sparse_morton_order')
        return order
    c = coors.long()
    key = (c[:, 0] << 48) | (_spread3(c[:, 1]) << 2) | (_spread3(c[:, 2]) << 1) | _spread3(c[:, 3])
    return torch.argsort(key)


class _Halo:
    """Tiling of a submanifold rule book for gga_sparse_conv_apply_halo (include/gga_hip.h): tiles of 256 rows in Z-order,
    each with the list of distinct input rows its entries name and the entries rewritten as positions in that list
    (gga_sparse_halo_build)."""

    def __init__(self, coors, rb):
        L = _lib.lib()
        TM = int(L.gga_sparse_halo_tile_rows())
        kvol, n = rb.nbr.shape
        dev = rb.nbr.device
        self.n_tiles = T = (n + TM - 1) // TM
        self.tile_rows = torch.full((T * TM,), -1, dtype=torch.int32, device=dev)
        self.tile_rows[:n] = morton_order(coors, *(rb.extent or (None, None))).int()
        self.capacity = kvol * TM
        self.halo_rows = torch.empty((T, self.capacity), dtype=torch.int32, device=dev)
        self.counts = torch.empty(T, dtype=torch.int32, device=dev)
        self.local_map = torch.empty((T, kvol, TM), dtype=torch.int16, device=dev)
        check(L.gga_sparse_halo_build(F._p(rb.nbr), F._p(self.tile_rows), n, T, kvol, self.capacity, F._p(self.halo_rows),
                                      F._p(self.counts), F._p(self.local_map), F._stream()), 'gga_sparse_halo_build')


class _Level:
    """One resolution level: coordinates + hash index + cached rule books."""

    def __init__(self, coors, spatial_shape, batch_size, index=None, index_n=None):
        self.coors = coors.contiguous()
        self.n = int(coors.shape[0])
        self.shape = tuple(int(s) for s in spatial_shape)
        self.batch_size = int(batch_size)
        self.index, self.index_n = index, index_n
        self._subm = {}

    def ensure_index(self):
        if self.index is None:
            L = _lib.lib()
            self.index_n = self.n
            self.index = torch.empty(L.gga_sparse_index_bytes(self.n), dtype=torch.uint8, device=self.coors.device)
            D, H, W = self.shape
            check(L.gga_sparse_build_index(F._p(self.coors), self.n, self.batch_size, D, H, W, F._p(self.index),
                                           self.index.numel(), F._stream()), 'gga_sparse_build_index')
        return self.index

    def subm_rulebook(self, kernel):
        nbr = self._subm.get(kernel)
        if nbr is None:
            self.ensure_index()
            kvol = kernel[0] * kernel[1] * kernel[2]
            nbr = torch.empty((kvol, self.n), dtype=torch.int32, device=self.coors.device)
            pad = tuple(k // 2 for k in kernel)
            check(_lib.lib().gga_sparse_rulebook(F._p(self.coors), self.n, F._p(self.coors), self.n, self.batch_size,
                                                 _i3(self.shape), _i3(self.shape), _i3(kernel), _i3((1, 1, 1)), _i3(pad),
                                                 F._p(self.index), self.index_n, None, 0, F._p(nbr), None,
                                                 F._stream()), 'gga_sparse_rulebook')
            cells = self.batch_size * self.shape[0] * self.shape[1] * self.shape[2]
            nbr = _Rulebook(nbr, self.coors, self.n / max(cells, 1), (self.batch_size, max(self.shape)))
            self._subm[kernel] = nbr
        return nbr

    def strided(self, kernel, stride, padding):
        """-> (output level, nbr [kvol,n_out], nbr_t [kvol,n_in])"""
        L = _lib.lib()
        dev = self.coors.device
        self.ensure_index()
        kvol = kernel[0] * kernel[1] * kernel[2]
        cap = self.n * min(kvol, 8)       # a site reaches at most ceil(k/s)^3 <= 8 outputs for k=3, s>=2
        for a in range(3):
            cap = cap if stride[a] >= 2 or kernel[a] == 1 else self.n * kvol
        out_coors = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        n_out_d = torch.zeros(1, dtype=torch.int32, device=dev)
        out_index = torch.empty(L.gga_sparse_out_index_bytes(self.n, kvol), dtype=torch.uint8, device=dev)
        ws = F._workspace('sp_sites', L.gga_sparse_out_sites_workspace_bytes(self.n, kvol), dev)
        out_dhw = (C.c_int32 * 3)()
        check(L.gga_sparse_conv_out_sites(F._p(self.coors), self.n, self.batch_size, _i3(self.shape), _i3(kernel),
                                          _i3(stride), _i3(padding), out_dhw, F._p(out_coors), cap, F._p(n_out_d),
                                          F._p(out_index), out_index.numel(), F._p(ws), ws.numel(), F._stream()),
              'gga_sparse_conv_out_sites')
        n_out = int(n_out_d.item())        # data-dependent size: one 4-byte read back, as spconv does
        if not 0 <= n_out <= cap:
            raise RuntimeError(f'sparse conv output sites: count {n_out} outside [0, {cap}]')
        if n_out == 0:                     # e.g. a (3,1,1)/(2,1,1) conv over sites that all sit in an odd, unpadded top slice
            return _empty_level(tuple(out_dhw), self.batch_size, dev), None, None
        out = _Level(out_coors[:n_out], tuple(out_dhw), self.batch_size, index=out_index, index_n=self.n * kvol)
        nbr = torch.empty((kvol, n_out), dtype=torch.int32, device=dev)
        nbr_t = torch.empty((kvol, self.n), dtype=torch.int32, device=dev)
        check(L.gga_sparse_rulebook(F._p(out.coors), n_out, F._p(self.coors), self.n, self.batch_size, _i3(self.shape),
                                    _i3(out.shape), _i3(kernel), _i3(stride), _i3(padding), F._p(self.index),
                                    self.index_n, F._p(out_index), out.index_n, F._p(nbr), F._p(nbr_t), F._stream()),
              'gga_sparse_rulebook')
        return out, _Rulebook(nbr), _Rulebook(nbr_t)


def _empty_level(shape, batch_size, device):
    return _Level(torch.zeros((0, 4), dtype=torch.int32, device=device), shape, batch_size)


def conv_out_shape(shape, kernel, stride, padding):
    return tuple((shape[i] + 2 * padding[i] - kernel[i]) // stride[i] + 1 for i in range(3))


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, _level=None):
        F._need_cuda(features, indices)
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = list(spatial_shape)
        self.batch_size = batch_size
        self.indice_dict = {}
        self._level = _level or _Level(self.indices, spatial_shape, batch_size)

    def replace_feature(self, new_features):
        out = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, _level=self._level)
        out.indice_dict = self.indice_dict
        return out

    @property
    def spatial_size(self):
        return int(torch.tensor(self.spatial_shape).prod())

    def dense(self, channels_first=True):
        """[N,C] features -> dense [B, C, D, H, W] (zeros elsewhere): the pillar-scatter canvas kernel
        with (z,y) folded into the row index."""
        D, H, W = self._level.shape
        c = self.indices
        if c.shape[0] == 0:
            out = self.features.new_zeros((self.batch_size, self.features.shape[1], D, H, W)) + 0 * self.features.sum()
            return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()
        folded = torch.stack([c[:, 0], torch.zeros_like(c[:, 0]), c[:, 1] * H + c[:, 2], c[:, 3]], 1).contiguous()
        out = F.pillar_scatter(self.features, folded, self.batch_size, D * H, W, unique=True)
        out = out.view(self.batch_size, self.features.shape[1], D, H, W)
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()


class IndexPlan:
    """Index structures of one batch for a stack of sparse convolutions (levels, rule books), built
    ahead of the features by ``build_index_plan`` - e.g. on a side stream while the previous
    step's backward is still running, so that the host reads of the site counts do not drain the
    main stream's launch queue."""

    def __init__(self, level0, indice_dict):
        self.level0, self.indice_dict = level0, indice_dict

    def tensors(self):
        """Every device tensor the plan holds (for ``Tensor.record_stream`` when it was built on
        another stream than the one that consumes it)."""
        levels, books = [self.level0], []
        for lvl, out_lvl, rb, rb_t in self.indice_dict.values():
            levels.append(out_lvl)
            books += [rb, rb_t]
        for lvl in levels:
            books += list(lvl._subm.values())
            for t in (lvl.coors, lvl.index):
                if t is not None:
                    yield t
        for rb in books:
            if rb is not None:
                for t in (rb.nbr, rb.mask, rb.perm):
                    if t is not None:
                        yield t


def build_index_plan(module, coors, spatial_shape, batch_size):
    """Walk the ``SparseConvolution`` layers of ``module`` in execution order (= registration order
    for the sequential encoders of the reference) and build every level / rule book they will ask for."""
    coors = coors if coors.dtype == torch.int32 else coors.int()
    lvl = level0 = _Level(coors, spatial_shape, batch_size)
    indice_dict = {}
    for m in module.modules():
        if not isinstance(m, SparseConvolution):
            continue
        if m.subm:
            if lvl.n:
                lvl.subm_rulebook(m.kernel_size)
        else:
            cached = indice_dict.get(m.indice_key) if m.indice_key else None
            if cached is None or cached[0] is not lvl:
                if lvl.n == 0:
                    cached = (lvl, _empty_level(conv_out_shape(lvl.shape, m.kernel_size, m.stride, m.padding),
                                                lvl.batch_size, coors.device), None, None)
                else:
                    cached = (lvl,) + lvl.strided(m.kernel_size, m.stride, m.padding)
                if m.indice_key:
                    indice_dict[m.indice_key] = cached
            lvl = cached[1]
    return IndexPlan(level0, indice_dict)


class SparseModule(nn.Module):
    """Marker base: modules that take and return a SparseConvTensor."""


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], dict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(name or str(len(self._modules)), module)

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            i += 1
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.features.shape[0] == 0:
                    continue
                if isinstance(m, nn.modules.batchnorm._BatchNorm) and i < len(mods) and isinstance(mods[i], nn.ReLU):
                    x = x.replace_feature(F.bn_act(x.features, m, relu=True))     # fused BN + ReLU on [N, C]
                    i += 1
                else:
                    x = x.replace_feature(m(x.features))
            else:
                x = m(x)
        return x


# fp32 products through three bf16 planes per operand (nine exact partial products on the bf16
# matrix cores, fp32 accumulation): 1.7x the fp32 MFMA rate at no loss of accuracy
# (csrc/sparse_conv.hip, tools_dev/micro/bf16x9_probe.hip). False = the native fp32 MFMA kernel.
SPLIT_BF16 = True


# Halo form of the submanifold gather-GEMM (csrc/sparse_conv.hip::sp_conv_halo_kernel; two fp16 planes) for the output widths in
# HALO_COLUMNS: 128. (The entry point also takes 64 columns - tests/test_sparse_gpu.py keeps that covered - but the product does
# not send them there: half the matrix work per gathered byte, no difference in the step, profiles/r05_sp_halo2.txt.) It walks
# all kvol offsets of every row block where the default kernel skips the offsets none of a block's 32 mask-sorted rows uses, and
# is 7 % faster than the default kernel on the 128-channel level of the shipped config (0.36 occupancy, 14.5 of 27 offsets per row).
# Round 6: OFF by default (GGA_SP_HALO=1 turns it on). The default kernel now sums every offset's products as a chain of its own
# (sp_conv_x9_kernel, `offset_sums`) and is as close to float64 as a per-offset fp32 sgemm (RMS 0.7-1.2 x, tests/test_precision_gpu.py);
# the halo form walks chunk-outer with ONE accumulator chain of cin / 32 x 27 x 6 roundings per element (4.9 x that sgemm at 128
# channels), its 128 accumulators per wave leave no registers for a second set, and ending a chain through the accumulator file
# costs ~5 vector instructions per element (measured: the compiler spills 1 000 registers; by hand 640 instructions per chain end):
# 0.8 ms of the 52 ms step bought the last rows of the precision table (profiles/r06_sparse_chain_ab.txt).
HALO = int(os.environ.get('GGA_SP_HALO', '0'))
HALO_COLUMNS = (128,)
HALO_MIN_ROWS = int(os.environ.get('GGA_SP_HALO_MIN_ROWS', '65536'))
HALO_MIN_OCCUPANCY = float(os.environ.get('GGA_SP_HALO_MIN_OCCUPANCY', '0.25'))



def planes():
    """Arithmetic of the split-plane gather kernels: 2 = two fp16 planes of the scaled operands / three partial
    products, 3 = three bf16 planes / six (dense_conv.PLANES, one switch for all matrix kernels)."""
    from . import dense_conv
    return dense_conv.PLANES


def amax_bits(t):
    """Absmax bits of ``t``: left by its producer (the fused BatchNorm kernels) when possible, else one pass."""
    from . import dense_conv
    return dense_conv.tensor_amax(t)


def _pack_weight(w, kvol, cin, cout, transpose, split=None, w_amax=None):
    """[kvol,cin,cout] (or [kvol,cout,cin] with ``transpose``) -> the conv kernel's operand order
    (gga_sparse_pack_weight / gga_sparse_pack_weight_split). ``w_amax``: absmax bits of the weight -> two fp16
    planes; None -> three bf16 planes."""
    L = _lib.lib()
    if SPLIT_BF16 if split is None else split:
        wp = torch.empty(L.gga_sparse_split_weight_bytes(kvol, cin, cout) // 2, dtype=torch.int16, device=w.device)
        check(L.gga_sparse_pack_weight_planes(F._p(w), kvol, cin, cout, transpose, 2 if w_amax is not None else 3, F._p(w_amax),
                                              F._p(wp), F._stream()), 'gga_sparse_pack_weight_split')
        return wp
    wp = torch.empty(L.gga_sparse_packed_weight_bytes(kvol, cin, cout) // 4, dtype=torch.float32, device=w.device)
    check(L.gga_sparse_pack_weight(F._p(w), kvol, cin, cout, transpose, F._p(wp), F._stream()), 'gga_sparse_pack_weight')
    return wp


def _conv_apply(x, rb, wp, n_rows, kvol, cin, cout, flip, y, x_amax=None, w_amax=None, stats=None, bn=None):
    """``x_amax`` / ``w_amax`` given: ``wp`` holds two fp16 planes (packed with the same ``w_amax``). ``stats``: f64
    [gga_sparse_conv_apply_tiles(n_rows), 2, cout] for the per-channel sums of y (split-plane kernels only). ``bn``
    (backward-data launches, with ``stats``): ``BnSource.part`` pointers of the BatchNorm + ReLU whose output gradient y
    is - y is stored masked by the ReLU and ``stats`` receives that BatchNorm's backward sums."""
    L = _lib.lib()
    if (HALO and wp.dtype == torch.int16 and w_amax is not None and rb.coors is not None and cin % 32 == 0 and 9 <= kvol <= 27
            and cout in HALO_COLUMNS and n_rows >= HALO_MIN_ROWS and rb.occupancy >= HALO_MIN_OCCUPANCY):
        hl = rb.halo()
        check(L.gga_sparse_conv_apply_halo(F._p(x), F._p(wp), F._p(hl.tile_rows), F._p(hl.counts), hl.capacity, F._p(hl.halo_rows),
                                           F._p(hl.local_map), n_rows, hl.n_tiles, kvol, cin, cout, flip, F._p(y), cout, 2, F._p(x_amax),
                                           F._p(w_amax), F._p(stats), *(bn if bn else (None, 0, None, None, None, None)), F._stream()),
              'gga_sparse_conv_apply_halo')
        return
    if wp.dtype == torch.int16:
        check(L.gga_sparse_conv_apply_bn_bwd(F._p(x), F._p(rb.nbr), F._p(wp), F._p(rb.perm), F._p(rb.mask), n_rows,
                                             kvol, cin, cout, flip, F._p(y), cout, 2 if w_amax is not None else 3, F._p(x_amax),
                                             F._p(w_amax), F._p(stats), *(bn if bn else (None, 0, None, None, None, None)),
                                             F._stream()), 'gga_sparse_conv_apply_split')
    else:
        check(L.gga_sparse_conv_apply(F._p(x), F._p(rb.nbr), F._p(wp), F._p(rb.perm), F._p(rb.mask), n_rows, kvol, cin, cout,
                                      flip, F._p(y), F._stream()), 'gga_sparse_conv_apply')


def conv_wgrad(x, gy, nbr, n_rows, kvol, cin, cout, gw, x_amax=None, g_amax=None):
    """gw [kvol,cin,cout] = sum over the pairs of ``nbr`` of x[in]^T gy[out]: the deterministic
    split-plane kernel (default; two fp16 planes when the operands' absmax bits are given, three bf16 planes
    otherwise) or the fp32-MFMA kernel with float atomics (``SPLIT_BF16 = False``)."""
    L = _lib.lib()
    if SPLIT_BF16:
        ws = F._workspace('sp_wgrad', L.gga_sparse_conv_wgrad_workspace_bytes(n_rows, kvol, cin, cout), x.device)
        two = x_amax is not None and g_amax is not None
        check(L.gga_sparse_conv_wgrad_planes(F._p(x), cin, F._p(gy), cout, F._p(nbr), n_rows, kvol, cin, cout, F._p(gw),
                                             2 if two else 3, F._p(x_amax) if two else None, F._p(g_amax) if two else None,
                                             F._p(ws), ws.numel(), F._stream()), 'gga_sparse_conv_wgrad_split')
    else:
        check(L.gga_sparse_conv_wgrad(F._p(x), F._p(gy), F._p(nbr), n_rows, kvol, cin, cout, F._p(gw), F._stream()),
              'gga_sparse_conv_wgrad')
    return gw


class _SparseConvFn(torch.autograd.Function):
    """features [n_in,Cin] x weight [kvol,Cin,Cout] -> [n_out,Cout] through rule book ``rb``
    (``rb_t`` = transposed rule book for the backward-data pass, None for submanifold convs)."""

    @staticmethod
    def forward(ctx, feats, weight, rb, rb_t, n_out):
        feats, w = feats.contiguous(), weight.contiguous()
        kvol = rb.nbr.shape[0]
        cin, cout = w.shape[-2], w.shape[-1]
        y = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
        two = SPLIT_BF16 and planes() == 2
        x_amax = amax_bits(feats) if two else None
        banked = SPLIT_BF16 and cout <= 128 and w.data_ptr() == weight.data_ptr()      # the parameter itself: its operands live in the weight bank
        w_amax = amax_bits(w.detach()) if two and not banked else None
        # the per-channel sums of y for the BatchNorm that follows (split-plane kernels; widths the fused BatchNorm takes)
        stats = None
        if SPLIT_BF16 and cout % 4 == 0 and cout <= 128 and n_out >= 1:
            stats = torch.empty((int(_lib.lib().gga_sparse_conv_apply_tiles(n_out)), 2, cout), dtype=torch.float64, device=feats.device)
        if banked:
            from . import dense_conv, weight_bank
            wp, w_amax = weight_bank.gather_operand(w.detach().reshape(kvol, cin, cout), planes())
            if two and dense_conv.RANGE_GUARD.armed:
                dense_conv.RANGE_GUARD.record(w.detach())
        else:
            wp = _pack_weight(w, kvol, cin, cout, 0, w_amax=w_amax)
        _conv_apply(feats, rb, wp, n_out, kvol, cin, cout, 0, y, x_amax, w_amax, stats)
        ctx.save_for_backward(feats, w)
        ctx.rb, ctx.rb_t, ctx.amax, ctx.banked = rb, rb_t, (x_amax, w_amax), banked
        # feats = relu(bn(.)) with this convolution as its consumer: the backward-data pass then does that BatchNorm's reduce
        from . import dense_conv
        ctx.bn_src = dense_conv.bn_source(feats, cin) if (SPLIT_BF16 and cin % 4 == 0) else None
        if stats is None:
            stats = torch.empty(0, dtype=torch.float64, device=feats.device)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return y, stats

    @staticmethod
    def backward(ctx, gy, _gstats=None):
        feats, w = ctx.saved_tensors
        rb, rb_t = ctx.rb, ctx.rb_t
        gy = gy.contiguous()
        kvol, n_out = rb.nbr.shape
        n_in = feats.shape[0]
        cin, cout = w.shape[-2], w.shape[-1]
        L = _lib.lib()
        gx = gw = None
        x_amax, w_amax = ctx.amax
        g_amax = amax_bits(gy) if x_amax is not None else None        # one pass for both consumers of gy
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(feats)
            if ctx.banked:                    # W[k]^T: the transposed view of the parameter, from the weight bank
                from . import weight_bank
                wt, w_amax = weight_bank.gather_operand(w.detach().reshape(kvol, cin, cout).transpose(1, 2), planes())
            else:
                wt = _pack_weight(w, kvol, cout, cin, 1, w_amax=w_amax)   # W[k]^T in fragment order
            tb, flip = (rb_t, 0) if rb_t is not None else (rb, 1)     # SubM: transposed map = reversed offsets
            src = ctx.bn_src
            if src is not None:
                from . import dense_conv
                st = torch.empty((int(L.gga_sparse_conv_apply_tiles(n_in)), 2, cin), dtype=torch.float64, device=gy.device)
                _conv_apply(gy, tb, wt, n_in, kvol, cout, cin, flip, gx, g_amax, w_amax, st, src.part(0, cin))
                gx._gga_bn_bwd = dense_conv.BnPartials(gx, [(0, cin, st)], tuple(p[5].data_ptr() for p in src.parts))
            else:
                _conv_apply(gy, tb, wt, n_in, kvol, cout, cin, flip, gx, g_amax, w_amax)
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            conv_wgrad(feats, gy, rb.nbr, n_out, kvol, cin, cout, gw, x_amax, g_amax)
        return gx, gw, None, None, None


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, indice_key=None, **kw):
        super().__init__()
        assert ndim == 3 and groups == 1 and _triple(dilation) == (1, 1, 1), 'only plain 3D sparse convs are on the GGA path'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.subm, self.indice_key = subm, indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # mmcv SparseConvolution.reset_parameters: kaiming_uniform_(a=sqrt(5)) on the 5-D weight
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    # Checkpoint layouts (mmdet3d/ops/spconv/overwrite_spconv/write_spconv2.py:42-101): mmcv's sparse
    # convs - and this module - keep the kernel as [kz, ky, kx, Cin, Cout]; spconv 2.x modules keep
    # [Cout, kz, ky, kx, Cin] and stamp their state_dict entries with version 2. The reference
    # permutes mmcv checkpoints on their way INTO spconv 2 modules; here the opposite direction is
    # needed: a checkpoint written by a reference run on spconv 2 loads into this module.
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + 'weight'
        w = state_dict.get(key)
        if w is not None and w.dim() == 5:
            mine = tuple(self.weight.shape)
            spconv2 = (mine[4],) + mine[:4]
            if (local_metadata.get('version', None) == 2 and tuple(w.shape) == spconv2) or \
                    (tuple(w.shape) == spconv2 and tuple(w.shape) != mine):
                state_dict = dict(state_dict)
                state_dict[key] = w.permute(1, 2, 3, 4, 0).contiguous()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def spconv2_weight(self):
        """The kernel in spconv 2.x layout [Cout, kz, ky, kx, Cin] (for exporting a checkpoint to a
        reference installation that runs on spconv 2)."""
        return self.weight.detach().permute(4, 0, 1, 2, 3).contiguous()

    def _run(self, feats, w, rb, rb_t, n_out):
        y, stats = _SparseConvFn.apply(feats, w, rb, rb_t, n_out)
        if stats.numel() and self.bias is None:
            F.attach_bn_partials(y, stats)       # per-channel sums of y for the BatchNorm that follows (functional.bn_act)
        return y

    def forward(self, x):
        assert isinstance(x, SparseConvTensor)
        lvl = x._level
        w = self.weight.view(-1, self.in_channels, self.out_channels)
        empty = lambda n: x.features.new_zeros((n, self.out_channels)) + 0 * w.sum()      # keeps the graph connected
        if self.subm:
            if lvl.n == 0:
                y = empty(0)
            else:
                y = self._run(x.features, w, lvl.subm_rulebook(self.kernel_size), None, lvl.n)
            out = x.replace_feature(y)
        else:
            cached = x.indice_dict.get(self.indice_key) if self.indice_key else None
            if cached is None or cached[0] is not lvl:
                if lvl.n == 0:       # no input site (an all-empty batch): an empty level of the right shape
                    cached = (lvl, _empty_level(conv_out_shape(lvl.shape, self.kernel_size, self.stride, self.padding),
                                                lvl.batch_size, x.features.device), None, None)
                else:
                    cached = (lvl,) + lvl.strided(self.kernel_size, self.stride, self.padding)
                if self.indice_key:
                    x.indice_dict[self.indice_key] = cached
            _, out_lvl, nbr, nbr_t = cached
            y = empty(0) if out_lvl.n == 0 else self._run(x.features, w, nbr, nbr_t, out_lvl.n)
            out = SparseConvTensor(y, out_lvl.coors, out_lvl.shape, x.batch_size, _level=out_lvl)
            out.indice_dict = x.indice_dict
        if self.bias is not None:
            out = out.replace_feature(out.features + self.bias)
        return out


@CONV_LAYERS.register_module()
class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


@CONV_LAYERS.register_module()
class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)
