"""Torch-facing wrappers over the C ABI (include/gga_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the current
HIP stream and autograd bookkeeping. Every op enqueues hand-written gfx950 kernels
from ``libgga_hip.so`` on ``torch.cuda.current_stream()`` through ctypes with raw
device pointers; nothing falls back to eager PyTorch or the CPU.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import LossParams, PfnParams, VoxelParams, check
import contextlib
import os

_DEFERRED_COUNTERS = None       # a list while a forward pass collects the BatchNorm counters it would increment


def count_batch(bn):
    """``num_batches_tracked += 1`` of a BatchNorm layer in training mode (torch/nn/modules/batchnorm.py,
    _BatchNorm.forward; the fused paths here all require ``momentum is not None``, so the counter is bookkeeping,
    not an input of the step). Inside :func:`deferred_batch_counters` the counters are only collected."""
    if _DEFERRED_COUNTERS is not None:
        _DEFERRED_COUNTERS.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


@contextlib.contextmanager
def deferred_batch_counters():
    """One multi-tensor add for all the BatchNorm counters of a forward pass instead of one one-element kernel per
    layer (36 per PointPillars step). A layer applied several times is counted that many times."""
    global _DEFERRED_COUNTERS
    prev, _DEFERRED_COUNTERS = _DEFERRED_COUNTERS, []
    try:
        yield
    except BaseException:
        _DEFERRED_COUNTERS = prev          # a pass that raised updated no running statistics for certain: count nothing
        raise
    seen, _DEFERRED_COUNTERS = _DEFERRED_COUNTERS, prev
    if seen:
        uniq = {}
        for t in seen:
            uniq.setdefault(id(t), [t, 0])[1] += 1
        torch._foreach_add_([t for t, _ in uniq.values()], [n for _, n in uniq.values()])

LAYOUT_NCHW, LAYOUT_NHWC = 0, 1

_workspaces = {}
_cell_maps = {}


def _stream():
    # raw handle of the current stream of the current device (what torch.cuda.current_stream().cuda_stream returns,
    # without building a Stream object: ~800 calls per train step)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('gga_amd ops run on the GPU only (got a CPU tensor); '
                               'there is no CPU fallback in the product path')


def _workspace(key, nbytes, device):
    """Grow-only scratch buffer per (purpose, device, stream)."""
    k = (key, device, torch.cuda.current_stream().cuda_stream)
    w = _workspaces.get(k)
    if w is None or w.numel() < nbytes:
        w = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[k] = w
    return w


# ----------------------------------------------------------------------------- a1/a2
def voxel_params(voxel_size, point_cloud_range, max_points, max_voxels):
    prm = VoxelParams()
    prm.voxel_size[:] = [float(v) for v in voxel_size]
    prm.pc_range[:] = [float(v) for v in point_cloud_range]
    prm.max_points, prm.max_voxels = int(max_points), int(max_voxels)
    return prm


def voxel_grid_size(prm):
    g = (C.c_int32 * 3)()
    _lib.lib().gga_voxel_grid_size(C.byref(prm), C.byref(g))
    return list(g)


@torch.no_grad()
def hard_voxelize_batch(points, voxel_size, point_cloud_range, max_points, max_voxels, sync=True):
    """Batched hard voxelization (mvx_two_stage_gga.py:211-236 in one call).

    points: list of [N_b, C] f32 CUDA tensors. Returns ``voxels [M,P,C]``,
    ``num_points [M]``, ``coors [M,4] (b,z,y,x)``, ``voxel_num [B+1]`` (device).
    With ``sync=True`` the outputs are trimmed to the exact M (one 4-byte D2H read, the
    reference syncs once per frame); with ``sync=False`` they keep the capacity
    ``B*max_voxels`` (rows >= M zero) and ``voxel_num[-1]`` carries M on the device.
    """
    _need_cuda(*points)
    B = len(points)
    ndim = points[0].shape[1]
    dev = points[0].device
    offs = np.zeros(B + 1, np.int64)
    offs[1:] = np.cumsum([p.shape[0] for p in points])
    cat = points[0].contiguous() if B == 1 else torch.cat(points, 0)
    if cat.dtype != torch.float32:
        cat = cat.float()
    total = int(offs[-1])
    prm = voxel_params(voxel_size, point_cloud_range, max_points, max_voxels)
    cap = B * int(max_voxels)
    voxels = torch.empty((cap, int(max_points), ndim), dtype=torch.float32, device=dev)
    coors = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num_points = torch.empty((cap,), dtype=torch.int32, device=dev)
    voxel_num = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    wsb = L.gga_hard_voxelize_workspace_bytes(B, total)
    ws = _workspace('vox', wsb, dev)
    check(L.gga_hard_voxelize_batch(_p(cat), ndim, offs.ctypes.data_as(C.POINTER(C.c_int64)), B,
                                    C.byref(prm), _p(voxels), _p(coors), _p(num_points), _p(voxel_num),
                                    _p(ws), ws.numel(), _stream()), 'gga_hard_voxelize_batch')
    if sync:
        m = int(voxel_num[-1].item())
        return voxels[:m], num_points[:m], coors[:m], voxel_num
    coors.num_valid = voxel_num[B:]           # device count travels with the coordinates (see num_valid_of)
    return voxels, num_points, coors, voxel_num


def num_valid_of(coors):
    """Device i32 [1] count of the rows that exist in capacity-sized voxel buffers (set by the
    ``sync=False`` voxelizer calls), or None for exact-sized ones."""
    return getattr(coors, 'num_valid', None)


class PreparedPoints:
    """Frames at capacity offsets with device-side counts: the hand-over from
    ``points_prepare_batch`` to ``hard_voxelize_prepared``."""

    def __init__(self, points, capacity_offsets, counts):
        self.points, self.capacity_offsets, self.counts = points, capacity_offsets, counts

    def __len__(self):
        return len(self.capacity_offsets) - 1

    def to_list(self):
        """Per-frame tensors (synchronises: reads the counts back)."""
        n = self.counts.cpu().tolist()
        return [self.points[int(o):int(o) + k] for o, k in zip(self.capacity_offsets[:-1], n)]


def upload(host, device):
    """Host array / CPU tensor -> device without draining the launch queue: a copy from pageable
    memory blocks the host until everything queued before it has run (the loss targets are
    uploaded after the forward pass was queued, so the host lost its lead exactly where the
    many small loss kernels start); staged through the caching pinned allocator the copy is
    just another stream operation."""
    t = torch.from_numpy(host) if isinstance(host, np.ndarray) else host
    dev = torch.device(device)
    if dev.type == 'cuda' and t.device.type == 'cpu' and t.numel():
        t = t.pin_memory()
    return t.to(dev, non_blocking=True)


def upload_many(arrays, device, align=256):
    """Host arrays {name: ndarray} -> {name: device tensor} through ONE pinned staging buffer and ONE copy: every copy on the
    compute stream is ~15-25 us of stream time with the device idle (the head's nine target arrays after the forward pass: 0.46 ms
    per step in a kernel + memory-copy trace, tools_dev/trace_region.py), whatever its size. The tensors are views of one device
    byte buffer (256-byte aligned)."""
    dev = torch.device(device)
    items, total = [], 0
    for k, a in arrays.items():
        a = np.ascontiguousarray(a)
        items.append((k, a, total))
        total += -(-max(a.nbytes, 1) // align) * align
    stage = torch.empty(total, dtype=torch.uint8, pin_memory=dev.type == 'cuda')
    sv = stage.numpy()
    for k, a, o in items:
        if a.nbytes:
            sv[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    d = stage.to(dev, non_blocking=True)
    out = {}
    for k, a, o in items:
        dt = torch.from_numpy(np.empty(0, a.dtype)).dtype
        out[k] = d[o:o + a.nbytes].view(dt).view(a.shape)
    return out


_CONSTS = {}


def const_tensor(data, device, dtype=torch.float32):
    """Device tensor of a small host constant (nested lists / tuples / a numpy array), uploaded once per (value, device,
    dtype): ``x.new_tensor(list)`` inside the step is a copy from pageable memory, i.e. a host wait for everything queued
    so far (91 of them per PGD step kept the host 40 ms behind). The result is shared: do not write into it."""
    key = (data.tobytes() + str(data.shape).encode() if isinstance(data, np.ndarray) else repr(data), str(device), dtype)
    t = _CONSTS.get(key)
    if t is None:
        if len(_CONSTS) > 4096:
            _CONSTS.clear()
        t = _CONSTS[key] = upload(torch.as_tensor(np.asarray(data), dtype=dtype), device)
        if t.is_cuda:                      # once per constant: it may be read from another stream right away
            torch.cuda.current_stream(t.device).synchronize()
    return t


def _cat_rows(parts, ndim, dev, dtype=torch.float32):
    parts = [torch.as_tensor(p, dtype=dtype).reshape(-1, ndim) for p in parts]
    offs = np.zeros(len(parts) + 1, np.int64)
    offs[1:] = np.cumsum([p.shape[0] for p in parts])
    cat = torch.cat(parts, 0) if parts else torch.zeros((0, ndim), dtype=dtype)
    return cat.to(dev, non_blocking=True).contiguous(), offs


@torch.no_grad()
def points_prepare_batch(scene, sampled, centers, min_distance, point_cloud_range, shuffle_seeds, device):
    """Device-side tail of the point pipeline for a batch (``gga_points_prepare_batch``): per frame
    drop scene points within ``min_distance`` (BEV) of a pasted object's centre, put the pasted
    objects' points in front, apply the strict range filter, permute by seed (0 = keep order).
    scene / sampled: lists of [n, C] f32 (host or device), centers: list of [k, 2] f64.
    -> ``PreparedPoints`` (no synchronisation)."""
    dev = torch.device(device)
    B = len(scene)
    ndim = int(scene[0].shape[1])
    d_scene, o_scene = _cat_rows(scene, ndim, dev)
    d_samp, o_samp = _cat_rows(sampled, ndim, dev)
    d_ctr, o_ctr = _cat_rows([np.asarray(c, np.float64).reshape(-1, 2) for c in centers], 2, dev, torch.float64)
    _need_cuda(d_scene)
    cap = (o_scene + o_samp).astype(np.int64)
    total = int(cap[-1])
    rng = torch.as_tensor(np.asarray(point_cloud_range, np.float32)).to(dev)
    seeds = (C.c_uint64 * B)(*[int(s) & 0xFFFFFFFFFFFFFFFF for s in shuffle_seeds])
    out = torch.empty((max(total, 1), ndim), dtype=torch.float32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    max_rows = int(np.max(np.diff(cap))) if B else 0
    ws = _workspace('pprep', L.gga_points_prepare_workspace_bytes(B, total, max_rows), dev)
    as64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    with torch.cuda.device(dev):
        check(L.gga_points_prepare_batch(_p(d_scene), as64(o_scene), _p(d_samp), as64(o_samp), _p(d_ctr), as64(o_ctr), B, ndim,
                                         float(min_distance), _p(rng), seeds, _p(out), _p(counts), _p(ws), ws.numel(),
                                         _stream()), 'gga_points_prepare_batch')
    return PreparedPoints(out, cap, counts)


def hard_voxelize_prepared(prep, voxel_size, point_cloud_range, max_points, max_voxels, sync=True):
    """``hard_voxelize_batch`` on a ``PreparedPoints`` (frames at capacity offsets, counts on the device)."""
    _need_cuda(prep.points)
    B, ndim, dev = len(prep), prep.points.shape[1], prep.points.device
    prm = voxel_params(voxel_size, point_cloud_range, max_points, max_voxels)
    cap = B * int(max_voxels)
    voxels = torch.empty((cap, int(max_points), ndim), dtype=torch.float32, device=dev)
    coors = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num_points = torch.empty((cap,), dtype=torch.int32, device=dev)
    voxel_num = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = _workspace('vox', L.gga_hard_voxelize_workspace_bytes(B, int(prep.capacity_offsets[-1])), dev)
    offs = np.ascontiguousarray(prep.capacity_offsets, np.int64)
    check(L.gga_hard_voxelize_prepared(_p(prep.points), ndim, offs.ctypes.data_as(C.POINTER(C.c_int64)), _p(prep.counts), B,
                                       C.byref(prm), _p(voxels), _p(coors), _p(num_points), _p(voxel_num), _p(ws),
                                       ws.numel(), _stream()), 'gga_hard_voxelize_prepared')
    if sync:
        m = int(voxel_num[-1].item())
        return voxels[:m], num_points[:m], coors[:m], voxel_num
    coors.num_valid = voxel_num[B:]
    return voxels, num_points, coors, voxel_num


@torch.no_grad()
def voxel_mean(voxels, num_points, num_features):
    _need_cuda(voxels, num_points)
    voxels = voxels.contiguous()
    m, P, ndim = voxels.shape
    out = torch.empty((m, num_features), dtype=torch.float32, device=voxels.device)
    if m == 0:
        return out
    num_points = num_points.contiguous()     # keep every converted tensor alive until the launch is enqueued
    check(_lib.lib().gga_voxel_mean(_p(voxels), _p(num_points), m, P, ndim, num_features,
                                    _p(out), _stream()), 'gga_voxel_mean')
    return out



# ----------------------------------------------------------------------------- a2'
class _FusedPFN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, voxels, num_points, coors, weight, gamma, beta, running_mean, running_var, prm, num_valid=None):
        _need_cuda(voxels, num_points, coors, weight)
        voxels, coors = voxels.contiguous(), coors.contiguous()
        m, P, _ = voxels.shape
        dev = voxels.device
        L = _lib.lib()
        out = torch.empty((m, 64), dtype=torch.float32, device=dev)
        argmax = torch.empty((m, 64), dtype=torch.uint8, device=dev)
        saved = torch.empty(239, dtype=torch.float64, device=dev)
        ws = _workspace('pfn', L.gga_pfn_workspace_bytes(m), dev)
        w = weight.contiguous()
        check(L.gga_pfn_fwd(_p(voxels), _p(num_points), _p(coors), m, _p(num_valid), P, C.byref(prm), _p(w), _p(gamma),
                            _p(beta), _p(running_mean), _p(running_var), _p(out), _p(argmax), _p(saved),
                            _p(ws), ws.numel(), _stream()), 'gga_pfn_fwd')
        ctx.save_for_backward(voxels, num_points, coors, w, gamma, out, argmax, saved)
        ctx.prm, ctx.num_valid = prm, num_valid
        return out

    @staticmethod
    def backward(ctx, g):
        voxels, num_points, coors, w, gamma, out, argmax, saved = ctx.saved_tensors
        m, P, _ = voxels.shape
        L = _lib.lib()
        gw, gg, gb = torch.empty_like(w), torch.empty_like(gamma), torch.empty_like(gamma)
        ws = _workspace('pfn', L.gga_pfn_workspace_bytes(m), g.device)
        g = g.contiguous()
        check(L.gga_pfn_bwd(_p(voxels), _p(num_points), _p(coors), m, _p(ctx.num_valid), P, C.byref(ctx.prm), _p(w),
                            _p(gamma), _p(out), _p(argmax), _p(saved), _p(g), _p(gw), _p(gg), _p(gb), _p(ws),
                            ws.numel(), _stream()), 'gga_pfn_bwd')
        return None, None, None, gw, gg, gb, None, None, None, None


def pfn_params(voxel_size, offsets, eps, momentum, training):
    prm = PfnParams()
    prm.voxel_size[:] = [float(v) for v in voxel_size]
    prm.offsets[:] = [float(v) for v in offsets]
    prm.eps, prm.momentum = float(eps), float(momentum)
    prm.training, prm.in_features, prm.channels = int(bool(training)), 4, 64
    return prm


def fused_pfn(voxels, num_points, coors, weight, gamma, beta, running_mean, running_var, prm, num_valid=None):
    """Fused PillarFeatureNet: [M,P,4] -> [M,64] (running stats updated in place when training).
    ``num_valid``: device i32 count of the pillars that exist when the buffers are capacity-sized
    (``hard_voxelize_batch(sync=False)``); the output rows past it are zero."""
    return _FusedPFN.apply(voxels, num_points, coors, weight, gamma, beta, running_mean, running_var, prm, num_valid)


# ----------------------------------------------------------------------------- a3
def _cell_map(device, batch, ny, nx):
    k = (device, batch, ny, nx, torch.cuda.current_stream().cuda_stream)
    m = _cell_maps.get(k)
    if m is None:
        nbytes = _lib.lib().gga_pillar_scatter_map_bytes(batch, ny, nx)
        m = torch.full((nbytes // 4,), -1, dtype=torch.int32, device=device)
        _cell_maps[k] = m
    return m


class _PillarScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, coors, batch, ny, nx, layout, num_valid, unique=False):
        _need_cuda(feats, coors)
        feats = feats.contiguous()
        coors = coors.contiguous()
        if coors.dtype != torch.int32:
            coors = coors.int()
        m, Cc = feats.shape
        if layout == LAYOUT_NCHW:
            canvas = torch.empty((batch, Cc, ny, nx), dtype=torch.float32, device=feats.device)
        else:
            canvas = torch.empty((batch, Cc, ny, nx), dtype=torch.float32, device=feats.device,
                                 memory_format=torch.channels_last)
        # unique coordinates + NHWC need no winner map (fill + row copies)
        cmap = None if (unique and layout == LAYOUT_NHWC) else _cell_map(feats.device, batch, ny, nx)
        check(_lib.lib().gga_pillar_scatter_fwd(_p(feats), _p(coors), m, _p(num_valid), batch, Cc, ny, nx,
                                                layout, int(bool(unique)), _p(cmap), _p(canvas), _stream()),
              'gga_pillar_scatter_fwd')
        ctx.save_for_backward(coors, num_valid)
        ctx.geom = (m, batch, Cc, ny, nx, layout)
        return canvas

    @staticmethod
    def backward(ctx, grad):
        coors, num_valid = ctx.saved_tensors
        m, batch, Cc, ny, nx, layout = ctx.geom
        if layout == LAYOUT_NCHW:
            grad = grad.contiguous()
        else:
            grad = grad.contiguous(memory_format=torch.channels_last)
        gf = torch.empty((m, Cc), dtype=torch.float32, device=grad.device)
        check(_lib.lib().gga_pillar_scatter_bwd(_p(grad), _p(coors), m, _p(num_valid), batch, Cc, ny, nx,
                                                layout, _p(gf), _stream()), 'gga_pillar_scatter_bwd')
        return gf, None, None, None, None, None, None, None


class _SparseBEV(torch.autograd.Function):
    """[n, C] features of distinct sites (b, z, y, x) -> [B, C * D, H, W] in channels-last memory, channel c * D + z
    (gga_sparse_bev_nhwc_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, feats, coors, B, D, H, W):
        _need_cuda(feats, coors)
        feats, coors = feats.contiguous(), coors.contiguous()
        n, C_ = feats.shape
        out = torch.empty((B, H, W, C_ * D), dtype=torch.float32, device=feats.device)
        check(_lib.lib().gga_sparse_bev_nhwc_fwd(_p(feats), _p(coors), n, B, C_, D, H, W, _p(out), _stream()), 'gga_sparse_bev_nhwc_fwd')
        ctx.save_for_backward(coors)
        ctx.geom = (n, B, C_, D, H, W)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        coors, = ctx.saved_tensors
        n, B, C_, D, H, W = ctx.geom
        g = g.contiguous(memory_format=torch.channels_last)
        gf = torch.empty((n, C_), dtype=torch.float32, device=g.device)
        check(_lib.lib().gga_sparse_bev_nhwc_bwd(_p(g), _p(coors), n, B, C_, D, H, W, _p(gf), _stream()), 'gga_sparse_bev_nhwc_bwd')
        return gf, None, None, None, None, None


def sparse_bev_channels_last(feats, coors, batch_size, D, H, W):
    """``SparseConvTensor.dense().view(N, C * D, H, W)`` of middle_encoders/sparse_encoder.py:134-138 as a channels-last tensor,
    without the NCHW map and the two layout copies in between (f32 CUDA features with C % 4 == 0, int32 coors, distinct sites)."""
    return _SparseBEV.apply(feats, coors, int(batch_size), int(D), int(H), int(W))


def pillar_scatter(feats, coors, batch_size, ny, nx, channels_last=False, num_valid=None, unique=False):
    """[M,C] pillar features + coors (b,z,y,x) -> dense [B,C,ny,nx] canvas. ``unique``: the caller
    guarantees distinct (b,y,x) (voxelizer output, sparse sites); otherwise the highest row of a
    duplicated cell wins."""
    canvas = _PillarScatter.apply(feats, coors, int(batch_size), int(ny), int(nx),
                                  LAYOUT_NHWC if channels_last else LAYOUT_NCHW, num_valid, bool(unique))
    if unique and channels_last:
        # lets the consumer's backward work on the pillars instead of the canvas (pillar_conv.py)
        from .pillar_conv import PillarSupport
        canvas.pillar_support = PillarSupport(feats, coors, num_valid)
    return canvas


# ----------------------------------------------------------------------------- a6/a7
_patch_tables = {}


def gaussian_patch_table(max_radius, device):
    """Patches for radius 0..max_radius with the reference's f64 formula
    (mmdet3d/core/utils/gaussian.py:6-22, sigma = diameter / 6), cast to f32."""
    k = (max_radius, device)
    if k not in _patch_tables:
        vals, offs = [], [0]
        for r in range(max_radius + 1):
            d = 2 * r + 1
            sigma = d / 6
            y, x = np.ogrid[-r:r + 1, -r:r + 1]
            h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
            h[h < np.finfo(h.dtype).eps * h.max()] = 0
            vals.append(h.astype(np.float32).reshape(-1))
            offs.append(offs[-1] + d * d)
        _patch_tables[k] = (torch.from_numpy(np.concatenate(vals)).to(device),
                            torch.tensor(offs, dtype=torch.int32, device=device))
    return _patch_tables[k]


@torch.no_grad()
def heatmap_splat(objs, n_maps, H, W, device, max_radius=64):
    """objs: [n,4] int32 (map index, cx, cy, radius) host or device -> [n_maps,H,W] f32."""
    hm = torch.empty((n_maps, H, W), dtype=torch.float32, device=device)
    objs = torch.as_tensor(objs, dtype=torch.int32).reshape(-1, 4)
    if not objs.is_cuda and objs.numel() and int(objs[:, 3].max()) > max_radius:       # (a device tensor: the caller vouches for max_radius)
        max_radius = int(objs[:, 3].max())
    table, offs = gaussian_patch_table(max_radius, device)
    objs = upload(objs.contiguous(), device)
    check(_lib.lib().gga_heatmap_splat(_p(hm), n_maps, H, W, _p(objs), objs.shape[0], _p(table), _p(offs),
                                       max_radius, _stream()), 'gga_heatmap_splat')
    return hm


# ----------------------------------------------------------------------------- a8
class _GaussianFocal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, alpha, gamma, scale):
        _need_cuda(logits, target)
        logits, target = logits.contiguous(), target.contiguous()
        n = logits.numel()
        L = _lib.lib()
        ws = _workspace('focal', L.gga_focal_loss_workspace_bytes(n), logits.device)
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        check(L.gga_focal_loss_fwd(_p(logits), _p(target), n, alpha, gamma, scale, _p(out), _p(ws),
                                   ws.numel(), _stream()), 'gga_focal_loss_fwd')
        ctx.save_for_backward(logits, target, out)
        ctx.cfg = (alpha, gamma, scale)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_loss, _g_npos):
        if g_loss is None:
            return None, None, None, None, None
        logits, target, out = ctx.saved_tensors
        alpha, gamma, scale = ctx.cfg
        grad = torch.empty_like(logits)
        g = g_loss.contiguous().float()
        check(_lib.lib().gga_focal_loss_bwd(_p(logits), _p(target), logits.numel(), alpha, gamma, scale,
                                            _p(out), _p(g), _p(grad), _stream()), 'gga_focal_loss_bwd')
        return grad, None, None, None, None


def gaussian_focal_loss(logits, target, alpha=2.0, gamma=4.0, scale=1.0):
    """clip_sigmoid + GaussianFocalLoss(reduction='mean', avg_factor=max(num_pos,1)), times
    ``scale``. Takes the raw heat-map logits. Returns (loss, num_pos) device scalars."""
    return _GaussianFocal.apply(logits, target, float(alpha), float(gamma), float(scale))


# ----------------------------------------------------------------------------- a9
class _GatherPred(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reg, height, dim, rot, ind, mask):
        _need_cuda(reg, height, dim, rot, ind, mask)
        B, _, H, W = reg.shape
        K = ind.shape[1]
        pred = torch.empty((B, K, 8), dtype=torch.float32, device=reg.device)
        # NB: bind the NCHW copies to names. `_p(x.contiguous())` would free each temporary as soon
        # as its pointer is taken, and the next copy would be allocated over it (channels-last maps).
        reg, height, dim, rot = reg.contiguous(), height.contiguous(), dim.contiguous(), rot.contiguous()
        check(_lib.lib().gga_gather_pred_fwd(_p(reg), _p(height),
                                             _p(dim), _p(rot), _p(ind), B, K, H, W,
                                             _p(pred), _stream()), 'gga_gather_pred_fwd')
        ctx.save_for_backward(ind, mask)
        ctx.geom = (B, K, H, W)
        return pred

    @staticmethod
    def backward(ctx, g):
        ind, mask = ctx.saved_tensors
        B, K, H, W = ctx.geom
        dev = g.device
        # one allocation, four views one behind the other: the entry point then clears them with ONE memset
        flat = torch.empty(B * 8 * H * W, dtype=torch.float32, device=dev)
        hw = B * H * W
        g_reg, g_h = flat[:2 * hw].view(B, 2, H, W), flat[2 * hw:3 * hw].view(B, 1, H, W)
        g_dim, g_rot = flat[3 * hw:6 * hw].view(B, 3, H, W), flat[6 * hw:].view(B, 2, H, W)
        g = g.contiguous()
        check(_lib.lib().gga_gather_pred_bwd(_p(g), _p(ind), _p(mask), B, K, H, W, _p(g_reg),
                                             _p(g_h), _p(g_dim), _p(g_rot), _stream()), 'gga_gather_pred_bwd')
        return g_reg, g_h, g_dim, g_rot, None, None


def gather_pred(reg, height, dim, rot, ind, mask):
    """cat(reg,height,dim,rot) gathered at ``ind`` -> pred [B,K,8] (head:657-676)."""
    return _GatherPred.apply(reg, height, dim, rot, ind.contiguous(), mask.contiguous())


# ----------------------------------------------------------------------------- a10-a13
_loss_params_cache = {}


def loss_params(B, K, train_cfg, l1_loss_weight=0.25, w_bpl=0.3, w_srl=0.1, w_pal=0.1):
    """The POD the loss kernels take; built once per (config object, B, K, weights) - reading a Config costs more
    than the kernels' launches."""
    key = (id(train_cfg), int(B), int(K), float(l1_loss_weight), float(w_bpl), float(w_srl), float(w_pal))
    hit = _loss_params_cache.get(key)
    if hit is not None and hit[0] is train_cfg:
        return hit[1]
    prm = _build_loss_params(B, K, train_cfg, l1_loss_weight, w_bpl, w_srl, w_pal)
    _loss_params_cache[key] = (train_cfg, prm)
    return prm


def _build_loss_params(B, K, train_cfg, l1_loss_weight, w_bpl, w_srl, w_pal):
    prm = LossParams()
    prm.B, prm.K = int(B), int(K)
    prm.fm_w = int(train_cfg['grid_size'][0]) // int(train_cfg['out_size_factor'])
    prm.voxel_size[:] = [float(v) for v in train_cfg['voxel_size'][:2]]
    prm.out_size_factor = float(train_cfg['out_size_factor'])
    prm.pc_range[:] = [float(v) for v in train_cfg['point_cloud_range'][:2]]
    prm.code_weights[:] = [float(v) for v in train_cfg['code_weights'][:5]]
    prm.l1_loss_weight = float(l1_loss_weight)
    prm.w_bpl, prm.w_srl, prm.w_pal = float(w_bpl), float(w_srl), float(w_pal)
    return prm


class _BoxLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, ind, mask, anno, l2i, bmask, ibp_xy, ibp_off, ibp_slot, prm):
        _need_cuda(pred, ind, mask, anno, l2i, bmask)
        B, K = prm.B, prm.K
        dev = pred.device
        L = _lib.lib()
        losses = torch.empty(5, dtype=torch.float32, device=dev)
        flat = torch.empty(B * K * (5 * 8 + 12), dtype=torch.float32, device=dev)      # one allocation: the entry point clears both with one memset
        grad_pred, box_out = flat[:5 * B * K * 8].view(5, B, K, 8), flat[5 * B * K * 8:].view(B, K, 12)
        ws = _workspace('box', L.gga_box_losses_workspace_bytes(B, K), dev)
        n_obj = 0 if ibp_slot is None else int(ibp_slot.shape[0])
        pred = pred.contiguous()
        check(L.gga_box_losses_fwd(_p(pred), _p(ind), _p(mask), _p(anno), _p(l2i), _p(bmask),
                                   _p(ibp_xy), _p(ibp_off), _p(ibp_slot), n_obj, C.byref(prm), _p(losses),
                                   _p(box_out), _p(grad_pred), _p(ws), ws.numel(), _stream()),
              'gga_box_losses_fwd')
        ctx.save_for_backward(grad_pred)
        ctx.geom = (B, K)
        ctx.mark_non_differentiable(box_out)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return losses, box_out

    @staticmethod
    def backward(ctx, g_losses, _g_box):
        (grad_pred,) = ctx.saved_tensors
        B, K = ctx.geom
        out = torch.empty((B, K, 8), dtype=torch.float32, device=grad_pred.device)
        g_losses = g_losses.contiguous().float()
        check(_lib.lib().gga_box_losses_bwd(_p(grad_pred), _p(g_losses), B, K, _p(out),
                                            _stream()), 'gga_box_losses_bwd')
        return (out,) + (None,) * 9


_ZERO_SCALAR = {}


def _zero_scalar(dev):
    z = _ZERO_SCALAR.get(str(dev))
    if z is None:
        z = _ZERO_SCALAR[str(dev)] = torch.zeros((), dtype=torch.float32, device=dev)
    return z


class _BoxLossTerms(torch.autograd.Function):
    """_BoxLosses with the five terms as five scalar outputs: a head that puts two of them into the total loss and logs the
    other three gets no UnbindBackward (a zero fill per unused term and a stack, every step and task) - the terms nobody
    differentiates arrive here as None and the gradient vector is ONE stack over a cached zero."""

    @staticmethod
    def forward(ctx, *args):
        losses, box_out = _BoxLosses.forward(ctx, *args)
        return (*losses.unbind(0), box_out)

    @staticmethod
    def backward(ctx, g0, g1, g2, g3, g4, _g_box):
        gs = (g0, g1, g2, g3, g4)
        if all(g is None for g in gs):
            return (None,) * 10
        z = _zero_scalar(ctx.saved_tensors[0].device)
        g_losses = torch.stack([z if g is None else g.float().reshape(()) for g in gs])
        return _BoxLosses.backward(ctx, g_losses, None)


def box_loss_terms(pred, ind, mask, anno_box, lidar2img, bound_mask, ibp_xy, ibp_offsets, ibp_slot, prm):
    """``box_losses`` with the losses as a tuple of five scalars (bpl, srl, pal_min, pal_x, pal_y) instead of one [5] tensor."""
    out = _BoxLossTerms.apply(pred, ind.contiguous(), mask.contiguous(), anno_box.contiguous(), lidar2img.contiguous(),
                              bound_mask.contiguous(), ibp_xy, ibp_offsets, ibp_slot, prm)
    return out[:5], out[5]


def box_losses(pred, ind, mask, anno_box, lidar2img, bound_mask, ibp_xy, ibp_offsets, ibp_slot, prm):
    """GGA losses of one task. Returns ``losses [5]`` (bpl, srl, pal_min, pal_x, pal_y — the
    final dict values) and ``box_out [B,K,12]`` (rot, l, w, box2d[4], X, Y, p2c_min/x/y)."""
    return _BoxLosses.apply(pred, ind.contiguous(), mask.contiguous(), anno_box.contiguous(),
                            lidar2img.contiguous(), bound_mask.contiguous(), ibp_xy, ibp_offsets, ibp_slot, prm)


# ----------------------------------------------------------------------------- fused BN (+res) (+ReLU)
class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, eps, momentum, training, relu, rows, C,
                partials=None):
        L = _lib.lib()
        dev = x.device
        y = torch.empty_like(x)                       # same strides (channels-last stays channels-last)
        saved = torch.empty(2 * C, dtype=torch.float32, device=dev)
        bits = torch.empty(L.gga_bn_relu_mask_bytes(rows, C), dtype=torch.uint8, device=dev) if relu else None
        ws = _workspace('bn', L.gga_bn_relu_workspace_bytes(rows, C), dev)
        from . import dense_conv
        amax = dense_conv.new_amax(dev)              # the apply pass leaves max |y| for the convolution that reads y
        use = partials is not None and training      # the producer of x already reduced the per-channel sums
        check(L.gga_bn_relu_fwd_ex(_p(x), _p(residual), _p(gamma), _p(beta), _p(running_mean), _p(running_var), rows, C,
                                   eps, momentum, int(training), int(relu), _p(y), C, _p(bits), _p(saved),
                                   _p(partials) if use else None, int(partials.shape[0]) if use else 0, _p(amax), _p(ws),
                                   ws.numel(), _stream()), 'gga_bn_relu_fwd')
        ctx.save_for_backward(x, gamma, saved, bits)
        ctx.cfg = (rows, C, relu, residual is not None, bool(training))
        if amax is None:
            amax = torch.empty(0, dtype=torch.int32, device=dev)
        ctx.mark_non_differentiable(amax, saved)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return y, amax, saved

    @staticmethod
    def backward(ctx, gy, _gamax=None, _gsaved=None):
        x, gamma, saved, bits = ctx.saved_tensors
        rows, C, relu, has_res, training = ctx.cfg
        L = _lib.lib()
        from . import dense_conv
        done = getattr(gy, '_gga_bn_bwd', None)      # the convolution that produced gy masked it and reduced the sums
        partials = done.take(gy, saved.data_ptr(), 0, C) if (done is not None and relu and not has_res) else None
        # the incoming gradient must have the memory layout of x ([rows, C] row major)
        gy = gy.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else gy.contiguous()
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if has_res else None
        gg, gb = torch.empty(C, dtype=torch.float32, device=x.device), torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _workspace('bn', L.gga_bn_relu_workspace_bytes(rows, C), x.device)
        amax = dense_conv.new_amax(x.device)         # max |gx| for the convolution backward that reads gx
        if partials is not None:
            check(L.gga_bn_relu_bwd_partials(_p(gy), C, _p(x), _p(gamma), _p(saved), rows, C, int(training), _p(partials),
                                             int(partials.shape[0]), _p(gx), _p(gg), _p(gb), _p(amax), _p(ws), ws.numel(),
                                             _stream()), 'gga_bn_relu_bwd_partials')
            dense_conv.set_amax(gx, amax)
            return gx, None, gg, gb, None, None, None, None, None, None, None, None, None
        check(L.gga_bn_relu_bwd_ex(_p(gy), C, _p(x), _p(bits), _p(gamma), _p(saved), rows, C, int(relu), int(training), _p(gx),
                                   _p(gres), _p(gg), _p(gb), _p(amax), _p(ws), ws.numel(), _stream()), 'gga_bn_relu_bwd')
        dense_conv.set_amax(gx, amax)
        return gx, gres, gg, gb, None, None, None, None, None, None, None, None, None


def attach_bn_partials(y, stats):
    """Remember the per-channel sums a convolution kernel left for its output ``y`` (valid while ``y`` is not written again)."""
    y.bn_partials = stats
    y._bn_partials_version = y._version
    return y


def bn_partials_of(x):
    """The producer's per-channel sums of ``x`` ([tiles, 2, C] f64) if ``x`` still holds what the producer wrote, else None."""
    p = getattr(x, 'bn_partials', None)
    if p is None or getattr(x, '_bn_partials_version', x._version) != x._version:
        return None
    return p if (p.dim() == 3 and p.shape[0] >= 1 and p.shape[2] == x.shape[1]) else None


def channel_sums(x):
    """``x.sum`` over every dimension but the channels of a [rows, C] / channels-last [B, C, H, W] gradient (a convolution's
    bias gradient) - ``gga_column_sums`` where the layout allows it, the framework's reduction otherwise."""
    rc = _rows_channels(x) if (x.is_cuda and x.dtype == torch.float32) else None
    C = x.shape[1]
    if rc is None or rc[0] < 1 or C % 4 or C // 4 > 256 or 256 % (C // 4) or x.data_ptr() % 16:
        return x.sum(tuple(d for d in range(x.dim()) if d != 1))
    L = _lib.lib()
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = _workspace('bn', L.gga_bn_relu_workspace_bytes(rc[0], C), x.device)
    check(L.gga_column_sums(_p(x), rc[0], C, _p(out), _p(ws), ws.numel(), _stream()), 'gga_column_sums')
    return out


def _rows_channels(x):
    """[rows, C] view of a 2-D contiguous or 4-D channels-last tensor, else None."""
    if x.dim() == 2 and x.is_contiguous():
        return x.shape[0], x.shape[1]
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not (x.shape[1] == 1):
        return x.shape[0] * x.shape[2] * x.shape[3], x.shape[1]
    return None


def bn_act(x, bn, relu=True, residual=None):
    """``relu(bn(x) + residual)`` (each part optional) — one fused HIP pass pair when ``x`` is a CUDA
    f32 [rows, C] / channels-last tensor: training mode (batch statistics) or evaluation mode (running statistics,
    also under autograd: a frozen ``norm_eval`` backbone); the eager ops otherwise."""
    rc = _rows_channels(x) if (x.is_cuda and x.dtype == torch.float32) else None
    C = x.shape[1]
    ok = (rc is not None and rc[0] >= 1 and C % 4 == 0 and C // 4 <= 256 and 256 % (C // 4) == 0 and bn.affine
          and bn.track_running_stats and bn.momentum is not None
          and (residual is None or (residual.shape == x.shape and residual.stride() == x.stride())))
    if not ok:
        y = bn(x)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    if bn.training:
        count_batch(bn)
    partials = bn_partials_of(x) if bn.training else None
    y, amax, saved = _BNAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                                  float(bn.momentum), bool(bn.training), bool(relu), int(rc[0]), int(C), partials)
    from . import dense_conv
    if amax.numel():
        dense_conv.set_amax(y, amax)
    if relu and residual is None and x.dim() in (2, 4) and y.requires_grad:
        # a 3x3 convolution that consumes y reduces this BatchNorm's backward sums in its backward-data epilogue
        y._gga_bn_src = dense_conv.BnSource(y, [(0, C, x.detach(), bn.weight.detach(), bn.bias.detach(), saved)])
    return y


class _GNAct(torch.autograd.Function):
    """``relu(group_norm(x))`` on a channels-last activation (gga_gn_relu_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        from . import dense_conv
        L = _lib.lib()
        B, C, H, W = x.shape
        dev = x.device
        y = torch.empty_like(x)
        stat = torch.empty((B, groups, 2), dtype=torch.float32, device=dev)
        ss = torch.empty((B, 2, C), dtype=torch.float32, device=dev)
        ws = _workspace('gn', L.gga_gn_relu_workspace_bytes(B, C), dev)
        amax = dense_conv.new_amax(dev)
        check(L.gga_gn_relu_fwd(_p(x), _p(gamma), _p(beta), B, H * W, C, groups, eps, int(relu), _p(y), _p(stat), _p(ss), _p(amax),
                                _p(ws), ws.numel(), _stream()), 'gga_gn_relu_fwd')
        ctx.save_for_backward(x, gamma, stat, ss)
        ctx.cfg = (groups, relu)
        if amax is None:
            amax = torch.empty(0, dtype=torch.int32, device=dev)
        ctx.mark_non_differentiable(amax)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return y, amax

    @staticmethod
    def backward(ctx, gy, _gamax=None):
        from . import dense_conv
        x, gamma, stat, ss = ctx.saved_tensors
        groups, relu = ctx.cfg
        L = _lib.lib()
        B, C, H, W = x.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x)
        gg = torch.empty(C, dtype=torch.float32, device=x.device)
        gb = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _workspace('gn', L.gga_gn_relu_workspace_bytes(B, C), x.device)
        amax = dense_conv.new_amax(x.device)
        check(L.gga_gn_relu_bwd(_p(gy), _p(x), _p(gamma), _p(stat), _p(ss), B, H * W, C, groups, int(relu), _p(gx), _p(gg), _p(gb),
                                _p(amax), _p(ws), ws.numel(), _stream()), 'gga_gn_relu_bwd')
        dense_conv.set_amax(gx, amax)
        return gx, gg, gb, None, None, None


def gn_act(x, gn, relu=True):
    """``relu(gn(x))`` - one fused HIP pass pair when ``x`` is a CUDA f32 channels-last [B, C, H, W] tensor whose group
    size is a multiple of 4 channels, the eager ops otherwise (``torch.nn.GroupNorm`` returns NCHW memory)."""
    C = x.shape[1] if x.dim() == 4 else 0
    ok = (x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
          and C % 4 == 0 and C // 4 <= 256 and 256 % (C // 4) == 0 and C % gn.num_groups == 0
          and (C // gn.num_groups) % 4 == 0 and gn.affine and x.shape[2] * x.shape[3] >= 1)
    if not ok:
        y = gn(x)
        return torch.relu(y) if relu else y
    y, amax = _GNAct.apply(x, gn.weight, gn.bias, int(gn.num_groups), float(gn.eps), bool(relu))
    if amax.numel():
        from . import dense_conv
        dense_conv.set_amax(y, amax)
    return y


class _BNActCat(torch.autograd.Function):
    """``cat([relu(bn_i(x_i))], dim=1)`` for channels-last inputs of equal [B, ., H, W]: every
    branch writes its column block of the concatenated map and reads its block of the gradient in
    place (gga_bn_relu_fwd_strided / _bwd_strided)."""

    @staticmethod
    def forward(ctx, n, cfg, *args):
        xs, gammas, betas = args[:n], args[n:2 * n], args[2 * n:3 * n]
        rms, rvs = args[3 * n:4 * n], args[4 * n:5 * n]
        partials = args[5 * n:6 * n]                  # per branch: the producer's per-channel sums of x, or None
        L = _lib.lib()
        x0 = xs[0]
        dev = x0.device
        B, _, H, W = x0.shape
        rows = B * H * W
        chans = [int(x.shape[1]) for x in xs]
        tot = sum(chans)
        out = torch.empty((B, tot, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        saved_all, bits_all = [], []
        off = 0
        from . import dense_conv
        amax = dense_conv.new_amax(dev)               # max over all branches' outputs = max of the concatenated map
        for i in range(n):
            eps, momentum, training = cfg[i]
            C = chans[i]
            saved = torch.empty(2 * C, dtype=torch.float32, device=dev)
            bits = torch.empty(L.gga_bn_relu_mask_bytes(rows, C), dtype=torch.uint8, device=dev)
            ws = _workspace('bn', L.gga_bn_relu_workspace_bytes(rows, C), dev)
            pt = partials[i] if (training and partials[i] is not None) else None
            check(L.gga_bn_relu_fwd_ex(_p(xs[i]), None, _p(gammas[i]), _p(betas[i]), _p(rms[i]), _p(rvs[i]), rows, C,
                                       eps, momentum, int(training), 1, out.data_ptr() + 4 * off, tot, _p(bits),
                                       _p(saved), _p(pt), int(pt.shape[0]) if pt is not None else 0, _p(amax), _p(ws), ws.numel(),
                                       _stream()), 'gga_bn_relu_fwd_strided')
            saved_all.append(saved)
            bits_all.append(bits)
            off += C
        ctx.save_for_backward(*xs, *gammas, *saved_all, *bits_all)
        ctx.n, ctx.chans, ctx.rows = n, chans, rows
        if amax is None:
            amax = torch.empty(0, dtype=torch.int32, device=dev)
        ctx.mark_non_differentiable(amax, *saved_all)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return (out, amax, *saved_all)

    @staticmethod
    def backward(ctx, g, _gamax=None, *_gsaved):
        from . import dense_conv
        n, chans, rows = ctx.n, ctx.chans, ctx.rows
        t = ctx.saved_tensors
        xs, gammas, saved_all, bits_all = t[:n], t[n:2 * n], t[2 * n:3 * n], t[3 * n:4 * n]
        L = _lib.lib()
        done = getattr(g, '_gga_bn_bwd', None)       # the convolution that produced g masked it and reduced the sums
        if done is not None:
            offs = [sum(chans[:i]) for i in range(n)]
            given = [done.take(g, saved_all[i].data_ptr(), offs[i], chans[i]) for i in range(n)]
        else:
            given = [None] * n
        g = g.contiguous(memory_format=torch.channels_last)
        tot = sum(chans)
        gxs, ggs, gbs = [], [], []
        off = 0
        for i in range(n):
            C = chans[i]
            x = xs[i]
            gx = torch.empty_like(x)
            gg = torch.empty(C, dtype=torch.float32, device=x.device)
            gb = torch.empty(C, dtype=torch.float32, device=x.device)
            ws = _workspace('bn', L.gga_bn_relu_workspace_bytes(rows, C), x.device)
            amax = dense_conv.new_amax(x.device)
            if given[i] is not None:
                check(L.gga_bn_relu_bwd_partials(g.data_ptr() + 4 * off, tot, _p(x), _p(gammas[i]), _p(saved_all[i]), rows, C, 1,
                                                 _p(given[i]), int(given[i].shape[0]), _p(gx), _p(gg), _p(gb), _p(amax), _p(ws),
                                                 ws.numel(), _stream()), 'gga_bn_relu_bwd_partials')
            else:
                check(L.gga_bn_relu_bwd_ex(g.data_ptr() + 4 * off, tot, _p(x), _p(bits_all[i]), _p(gammas[i]),
                                           _p(saved_all[i]), rows, C, 1, 1, _p(gx), None, _p(gg), _p(gb), _p(amax), _p(ws),
                                           ws.numel(), _stream()), 'gga_bn_relu_bwd_strided')
            dense_conv.set_amax(gx, amax)
            gxs.append(gx), ggs.append(gg), gbs.append(gb)
            off += C
        return (None, None, *gxs, *ggs, *gbs) + (None,) * (3 * n)


def bn_relu_cat(xs, bns):
    """``torch.cat([relu(bn(x)) for x, bn in zip(xs, bns)], dim=1)`` without the concat copy (and
    without the split copies in backward) when every branch qualifies for the fused BatchNorm
    kernel; the plain composition otherwise."""
    def fusable(x, bn):
        rc = _rows_channels(x) if (x.is_cuda and x.dtype == torch.float32) else None
        C = x.shape[1]
        return (rc is not None and C % 4 == 0 and C // 4 <= 256 and 256 % (C // 4) == 0 and bn.affine
                and bn.track_running_stats and bn.momentum is not None and (bn.training or not torch.is_grad_enabled()))
    same = all(x.shape[0] == xs[0].shape[0] and x.shape[2:] == xs[0].shape[2:] for x in xs)
    if len(xs) < 2 or not same or not all(fusable(x, bn) for x, bn in zip(xs, bns)):
        return torch.cat([bn_act(x, bn, relu=True) for x, bn in zip(xs, bns)], dim=1) if len(xs) > 1 \
            else bn_act(xs[0], bns[0], relu=True)
    for bn in bns:
        if bn.training:
            count_batch(bn)
    n = len(xs)
    cfg = tuple((float(bn.eps), float(bn.momentum), bool(bn.training)) for bn in bns)
    # the producers' per-channel sums (dense_conv / strided_conv / sparse convolutions), where they left them
    out, amax, *saved = _BNActCat.apply(n, cfg, *xs, *[bn.weight for bn in bns], *[bn.bias for bn in bns],
                                        *[bn.running_mean for bn in bns], *[bn.running_var for bn in bns],
                                        *[bn_partials_of(x) for x in xs])
    from . import dense_conv
    if amax.numel():
        dense_conv.set_amax(out, amax)
    if out.requires_grad and all(bn.training for bn in bns):
        parts, off = [], 0
        for x, bn, sv in zip(xs, bns, saved):
            parts.append((off, int(x.shape[1]), x.detach(), bn.weight.detach(), bn.bias.detach(), sv))
            off += int(x.shape[1])
        out._gga_bn_src = dense_conv.BnSource(out, parts)
    return out


# ----------------------------------------------------------------------------- head output convs
class _HeadConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        B, C, H, W = x.shape
        cout = weight.shape[0]
        w = weight.contiguous()                       # logical [cout, 64, 3, 3], NCHW-contiguous
        y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
        check(_lib.lib().gga_head_conv3x3_fwd(_p(x), C, None, _p(w), _p(bias), B, H, W, C, cout, _p(y), _stream()),
              'gga_head_conv3x3_fwd')
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        B, C, H, W = x.shape
        cout = w.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # the input gradient writes the full [B,64,H,W] tensor: already bandwidth-bound in MIOpen
            gx = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gw = torch.empty_like(w)
            gb = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            L = _lib.lib()
            ws = _workspace('headconv', L.gga_head_conv3x3_workspace_bytes(cout), x.device)
            check(L.gga_head_conv3x3_wgrad(_p(x), C, None, _p(gy), B, H, W, C, cout, _p(gw), _p(gb), _p(ws), ws.numel(),
                                           _stream()), 'gga_head_conv3x3_wgrad')
        return gx, gw, gb


class _BnReluHeadConv3x3(torch.autograd.Function):
    """``conv(relu(bn(x)))`` for the tail of a head branch without materialising the normalised
    activation: batch statistics (gga_bn_stats), then the output conv applies scale/shift + ReLU
    while it loads x; backward = the weight gradient from x with the same on-load affine, and the
    BatchNorm backward with the ReLU mask recomputed from x and the conv's input gradient rebuilt from
    grad_y in registers (gga_head_tail_bwd: no backward-data tensor)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, weight, bias, eps, momentum, training, partials=None):
        L = _lib.lib()
        B, C, H, W = x.shape
        rows, cout = B * H * W, weight.shape[0]
        dev = x.device
        w = weight.contiguous()
        saved = torch.empty(2 * C, dtype=torch.float32, device=dev)
        ss = torch.empty(2 * C, dtype=torch.float32, device=dev)
        if partials is not None:                    # sums left by the convolution that produced x
            check(L.gga_bn_stats_partials(_p(gamma), _p(beta), _p(running_mean), _p(running_var), rows, C, eps, momentum,
                                          _p(saved), _p(ss), _p(partials), int(partials.shape[0]), _stream()),
                  'gga_bn_stats_partials')
        else:
            ws = _workspace('bn', L.gga_bn_relu_workspace_bytes(rows, C), dev)
            check(L.gga_bn_stats(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var), rows, C, eps, momentum,
                                 int(training), _p(saved), _p(ss), _p(ws), ws.numel(), _stream()), 'gga_bn_stats')
        y = torch.empty((B, cout, H, W), dtype=torch.float32, device=dev)
        check(L.gga_head_conv3x3_fwd(_p(x), C, _p(ss), _p(w), _p(bias), B, H, W, C, cout, _p(y), _stream()),
              'gga_head_conv3x3_fwd')
        ctx.save_for_backward(x, gamma, saved, ss, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, saved, ss, w = ctx.saved_tensors
        L = _lib.lib()
        gy = gy.contiguous()
        B, C, H, W = x.shape
        rows, cout = B * H * W, w.shape[0]
        dev = x.device
        gw = torch.empty_like(w)
        gb = torch.empty(cout, dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws = _workspace('headconv', L.gga_head_conv3x3_workspace_bytes(cout), dev)
        check(L.gga_head_conv3x3_wgrad(_p(x), C, _p(ss), _p(gy), B, H, W, C, cout, _p(gw), _p(gb), _p(ws), ws.numel(),
                                       _stream()), 'gga_head_conv3x3_wgrad')
        # BatchNorm + ReLU backward with the conv's input gradient rebuilt from gy on the fly (never stored)
        gx = torch.empty_like(x)
        gg = torch.empty(C, dtype=torch.float32, device=dev)
        gbeta = torch.empty(C, dtype=torch.float32, device=dev)
        wsb = _workspace('bn', L.gga_bn_relu_workspace_bytes(rows, C), dev)
        from . import dense_conv
        amax = dense_conv.new_amax(dev)
        check(L.gga_head_tail_bwd(_p(gy), _p(x), C, _p(ss), _p(gamma), _p(saved), _p(w), B, H, W, C, cout, _p(gx), C, _p(gg), _p(gbeta),
                                  _p(amax), _p(wsb), wsb.numel(), _stream()), 'gga_head_tail_bwd')
        dense_conv.set_amax(gx, amax)
        return gx, gg, gbeta, None, None, gw, gb, None, None, None, None


def bn_relu_head_conv3x3(x, bn, conv):
    """``conv(relu(bn(x)))`` (tail of a SeparateHead branch: the ConvModule's norm + activation
    and the output conv), fused when both halves qualify for their HIP kernels in training mode."""
    rc = _rows_channels(x) if (x.is_cuda and x.dtype == torch.float32) else None
    ok = (rc is not None and x.dim() == 4 and x.shape[1] == 64 and bn.affine and bn.track_running_stats
          and bn.momentum is not None and bn.training and torch.is_grad_enabled() and conv.out_channels <= 4
          and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
          and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros')
    if not ok:
        return head_conv3x3(bn_act(x, bn, relu=True), conv)
    count_batch(bn)
    partials = bn_partials_of(x)
    return _BnReluHeadConv3x3.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.weight, conv.bias,
                                    float(bn.eps), float(bn.momentum), True, partials)


_BRANCH_STREAMS = {}
HEAD_STREAMS = int(os.environ.get('GGA_HEAD_STREAMS', '2'))      # streams the branch launches are dealt to (1: none but the caller's)


def _branch_streams(device, n):
    """Streams for the per-branch launches of a head (see _HeadBranches), the caller's first: every one of those launches is a
    persistent grid whose last round leaves most of the chip idle (2240 tiles on 512 workgroup slots: 4.4 rounds), and the
    branches are independent - dealt to two streams, one branch's tail runs beside the next branch's start."""
    k = max(1, min(HEAD_STREAMS, n))
    pool = _BRANCH_STREAMS.setdefault(device, [])
    while len(pool) < k - 1:
        pool.append(torch.cuda.Stream(device=device))
    return [torch.cuda.current_stream(device)] + pool[:k - 1]


class _HeadBranches(torch.autograd.Function):
    """All branches ``conv3x3(64 -> c_i)(relu(bn_i(conv3x3(64 -> 64)_i(x))))`` of a CenterHead (every task's
    SeparateHead, centerpoint_head.py:46-79) on the one shared feature map, as one autograd node.

    Forward: per branch the bf16x6 convolution writes its 64 channels (and their BatchNorm sums) into a column
    block of ONE [B, 64n, H, W] channels-last buffer, the statistics are folded, and the output conv normalises
    while it loads. Backward: per branch the output conv's weight gradient and ``gga_head_tail_bwd``, which
    writes the gradient w.r.t. the branch's column block of a second [B, 64n, H, W] buffer; then ONE
    backward-data convolution 64n -> 64 and ONE weight-gradient call over all branches - where one node per
    branch left autograd n - 1 full-size additions of the shared map's gradient."""

    @staticmethod
    def forward(ctx, x, n, cfg, *t):
        from . import dense_conv
        w1, gam, bet, rm, rv, w2, b2 = (t[i * n:(i + 1) * n] for i in range(7))
        L = _lib.lib()
        B, C, H, W = x.shape
        rows, dev, tot = B * H * W, x.device, C * n
        tr = dense_conv._transposed(H, W)
        Y = torch.empty((B, tot, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        outs, saved_all, ss_all = [], [], []
        x_amax = dense_conv.tensor_amax(x) if dense_conv.PLANES == 2 else None     # from x's producer, for all n convolutions
        # the first convolutions two branches at a time (64 -> 128 channels into adjacent column blocks of Y): the
        # 128-channel form of the kernel stages and splits the shared input once for both
        pair_stats = {}
        for i in range(0, n - 1, 2):          # (the pair's operand is assembled from the two parameters by the weight bank: no cat)
            _, st = dense_conv._run(x, [w1[i].detach(), w1[i + 1].detach()], False, True, x_amax, None, Y, C * i)
            pair_stats[i], pair_stats[i + 1] = (st, 0), (st, C)          # columns of the pair's [tiles, 2, 2C] rows: no copies
        all_stats = []
        for i in range(n):
            if i in pair_stats:
                all_stats.append(pair_stats[i])
            else:
                all_stats.append((dense_conv._run(x, w1[i].detach(), False, True, x_amax, None, Y, C * i)[1], 0))
        # everything the branch launches write is allocated here, on this stream; odd branches are LAUNCHED on a second one
        w2c = [w.contiguous() for w in w2]
        for i in range(n):
            outs.append(torch.empty((B, w2[i].shape[0], H, W), dtype=torch.float32, device=dev))
            saved_all.append(torch.empty(2 * C, dtype=torch.float32, device=dev))
            ss_all.append(torch.empty(2 * C, dtype=torch.float32, device=dev))
        streams = _branch_streams(dev, n)
        for st_ in streams[1:]:
            st_.wait_stream(streams[0])
        for i in range(n):
            eps, momentum = cfg[i]
            stats, col = all_stats[i]
            tiles = int(stats.shape[0])
            with torch.cuda.stream(streams[i % len(streams)]):
                check(L.gga_bn_stats_partials_cols(_p(gam[i]), _p(bet[i]), _p(rm[i]), _p(rv[i]), rows, C, eps, momentum,
                                                   _p(saved_all[i]), _p(ss_all[i]), _p(stats), tiles, int(stats.shape[2]), col,
                                                   _stream()), 'gga_bn_stats_partials_cols')
                check(L.gga_head_conv3x3_fwd(Y.data_ptr() + 4 * C * i, tot, _p(ss_all[i]), _p(w2c[i]), _p(b2[i]), B, H, W, C,
                                             w2[i].shape[0], _p(outs[i]), _stream()), 'gga_head_conv3x3_fwd')
        for st_ in streams[1:]:
            streams[0].wait_stream(st_)
        ctx.save_for_backward(x, Y, *w1, *gam, *w2, *saved_all, *ss_all)
        ctx.n, ctx.has_bias, ctx.x_amax = n, [b is not None for b in b2], x_amax
        ctx.bn_src = dense_conv.bn_source(x, C) if dense_conv.BN_BWD_FUSED else None     # x = relu(bn(shared conv))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        from . import dense_conv
        n = ctx.n
        t = ctx.saved_tensors
        x, Y = t[0], t[1]
        w1, gam, w2, saved_all, ss_all = (t[2 + i * n:2 + (i + 1) * n] for i in range(5))
        L = _lib.lib()
        B, C, H, W = x.shape
        rows, dev, tot = B * H * W, x.device, C * n
        G = torch.empty_like(Y)
        # the tail kernels leave max |G_i| of their column block: the weight gradient scales every branch's block by its own
        # maximum (a regression branch's gradient lives on a few object cells, orders of magnitude below a heat-map branch's:
        # under ONE scale for the 960 channels its two fp16 planes keep only a few bits - measured 3e-3 .. 7e-3 on those
        # branches' weight gradients where fp32 is at 1e-5); the backward-data convolution, whose result sums over the branches,
        # takes the largest of them
        g_blocks = dense_conv.new_amax(dev, n)
        gw2, gb2, ggam, gbet = [], [], [], []
        gyc = [g.contiguous() for g in gys]
        w2c = [w.contiguous() for w in w2]
        for i in range(n):                      # outputs of the branch launches: allocated on this stream (see forward)
            gw2.append(torch.empty_like(w2c[i]))
            gb2.append(torch.empty(w2[i].shape[0], dtype=torch.float32, device=dev) if ctx.has_bias[i] else None)
            ggam.append(torch.empty(C, dtype=torch.float32, device=dev))
            gbet.append(torch.empty(C, dtype=torch.float32, device=dev))
        streams = _branch_streams(dev, n)
        for st_ in streams[1:]:
            st_.wait_stream(streams[0])
        for i in range(n):
            cout = w2[i].shape[0]
            with torch.cuda.stream(streams[i % len(streams)]):
                ws = _workspace('headconv', L.gga_head_conv3x3_workspace_bytes(cout), dev)       # (scratch buffers are per stream)
                wsb = _workspace('bn', L.gga_bn_relu_workspace_bytes(rows, C), dev)
                check(L.gga_head_conv3x3_wgrad(Y.data_ptr() + 4 * C * i, tot, _p(ss_all[i]), _p(gyc[i]), B, H, W, C, cout, _p(gw2[i]),
                                               _p(gb2[i]), _p(ws), ws.numel(), _stream()), 'gga_head_conv3x3_wgrad')
                check(L.gga_head_tail_bwd(_p(gyc[i]), Y.data_ptr() + 4 * C * i, tot, _p(ss_all[i]), _p(gam[i]), _p(saved_all[i]), _p(w2c[i]),
                                          B, H, W, C, cout, G.data_ptr() + 4 * C * i, tot, _p(ggam[i]), _p(gbet[i]),
                                          _p(g_blocks[i:i + 1] if g_blocks is not None else None), _p(wsb), wsb.numel(),
                                          _stream()), 'gga_head_tail_bwd')
        for st_ in streams[1:]:
            streams[0].wait_stream(st_)
        wcat = [w.detach() for w in w1]            # stands for the [64n, 64, 3, 3] concatenation (weight bank: never made)
        g_amax = g_blocks.max().reshape(1) if g_blocks is not None else None       # (bits of non-negative floats order like ints)
        gx = dense_conv.run_bn_bwd(G, wcat, g_amax, None, ctx.bn_src) if ctx.needs_input_grad[0] else None
        gwcat = dense_conv._wgrad(x, G, wcat, ctx.x_amax if dense_conv.PLANES == 2 else None, g_blocks, g_per_block=True)
        gw1 = list(gwcat.split(C, dim=0))
        none = [None] * n
        return (gx, None, None, *gw1, *ggam, *gbet, *none, *none, *gw2, *gb2)


def head_branches(x, branches):
    """Outputs of the head branches ``[(conv1, bn, conv2), ...]`` that all read ``x`` (see _HeadBranches), or
    None when a branch does not qualify for the fused kernels (the caller then runs them one by one)."""
    from . import dense_conv
    rc = _rows_channels(x) if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4) else None
    if rc is None or x.shape[1] != 64 or not torch.is_grad_enabled() or not dense_conv.ENABLED or not dense_conv.WGRAD:
        return None
    for conv1, bn, conv2 in branches:
        ok = (dense_conv.eligible(conv1, x) and conv1.bias is None and conv1.out_channels == 64 and bn.affine
              and bn.track_running_stats and bn.momentum is not None and bn.training and conv2.out_channels <= 4
              and type(conv2) is torch.nn.Conv2d and conv2.in_channels == 64 and conv2.kernel_size == (3, 3)
              and conv2.stride == (1, 1) and conv2.padding == (1, 1) and conv2.dilation == (1, 1) and conv2.groups == 1
              and conv2.padding_mode == 'zeros')
        if not ok:
            return None
    for _, bn, _ in branches:
        count_batch(bn)
    n = len(branches)
    cfg = tuple((float(bn.eps), float(bn.momentum)) for _, bn, _ in branches)
    cols = ([c1.weight for c1, _, _ in branches], [bn.weight for _, bn, _ in branches], [bn.bias for _, bn, _ in branches],
            [bn.running_mean for _, bn, _ in branches], [bn.running_var for _, bn, _ in branches],
            [c2.weight for _, _, c2 in branches], [c2.bias for _, _, c2 in branches])
    return _HeadBranches.apply(x, n, cfg, *[t for col in cols for t in col])


def head_conv3x3(x, conv):
    """``conv(x)`` for the output convs of the head branches (64 -> 1..4 channels, 3x3, pad 1): the
    HBM-bound HIP kernels when ``x`` is a CUDA f32 channels-last activation, MIOpen otherwise."""
    ok = (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 64
          and x.is_contiguous(memory_format=torch.channels_last) and conv.out_channels <= 4
          and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
          and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros')
    if not ok:
        return conv(x)
    return _HeadConv3x3.apply(x, conv.weight, conv.bias)


def centerpoint_detect(boxes, scores, labels, coder_range, coder_score_threshold, score_threshold, limit_range, nms_threshold,
                       pre_max_size, post_max_size, num_classes):
    """``gga_centerpoint_detect``: boxes [T, B, K, D] / scores / labels [T, B, K] of all tasks' decoded top-k cells -> the
    detections of every frame (masks, rotated BEV NMS, range filter, task merge, bottom-centre z, global labels) in one launch.
    -> (out_boxes [B, T*K, D], out_scores [B, T*K], out_labels int32 [B, T*K], count int32 [B])."""
    _need_cuda(boxes, scores, labels)
    T, B, K, D = boxes.shape
    dev = boxes.device
    boxes, scores, labels = boxes.float().contiguous(), scores.float().contiguous(), labels.float().contiguous()
    out_boxes = torch.empty((B, T * K, D), dtype=torch.float32, device=dev)
    out_scores = torch.empty((B, T * K), dtype=torch.float32, device=dev)
    out_labels = torch.empty((B, T * K), dtype=torch.int32, device=dev)
    count = torch.zeros((B,), dtype=torch.int32, device=dev)
    offs, flag = [], 0
    for n in num_classes:
        offs.append(flag)
        flag += int(n)
    cr = const_tensor([float(v) for v in coder_range], dev)
    lr = const_tensor([float(v) for v in limit_range], dev) if limit_range is not None and len(limit_range) > 0 else None
    co = const_tensor(offs, dev, torch.int32)
    sc = const_tensor([int(int(n) == 1) for n in num_classes], dev, torch.int32)
    ws = _workspace('cp_detect', _lib.lib().gga_centerpoint_detect_workspace_bytes(T, B), dev)
    check(_lib.lib().gga_centerpoint_detect(_p(boxes), _p(scores), _p(labels), T, B, K, D, _p(cr),
                                            float(coder_score_threshold) if coder_score_threshold is not None else 0.0,
                                            int(coder_score_threshold is not None), float(score_threshold), _p(lr), float(nms_threshold),
                                            int(pre_max_size or 0), int(post_max_size or 0), _p(co), _p(sc), _p(out_boxes), _p(out_scores),
                                            _p(out_labels), _p(count), _p(ws), ws.numel(), _stream()), 'gga_centerpoint_detect')
    return out_boxes, out_scores, out_labels, count
