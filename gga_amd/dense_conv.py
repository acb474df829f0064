"""3x3 / stride-1 / pad-1 convolutions of the BEV trunk and the head branches on the repo's matrix kernels
(``gga_dense_conv3x3_planes`` / ``gga_dense_wgrad3x3_planes``): forward, backward-data and weight gradient; every other
eligible convolution is handed on to ``strided_conv`` (stride-2 3x3, 1x1, kernel = stride transposed).

Reference call sites: the block convolutions of ``SECOND`` (backbones/second.py:58-63) and the
``ConvModule`` that opens every ``SeparateHead`` branch (dense_heads/centerpoint_head.py:58-68).
fp32 in, fp32 out, fp32 accumulation: operands are split into 16-bit planes (``PLANES`` below: two fp16 planes of the
scaled operands and three partial products, or three bf16 planes and six) - error against float64 no larger than
MIOpen's fp32 kernels (tools_dev/bench_dense3x3.py, DESIGN.md 5 for the contract).
"""
import os

import torch
from torch import nn

from . import _lib
from . import functional as F
from ._lib import check

ENABLED = True          # False: every dense convolution goes to MIOpen
WGRAD = True            # False: weight gradients stay with MIOpen
# Arithmetic of the matrix kernels (both: fp32 in, fp32 out, fp32 accumulation):
#   3  three bf16 planes, six partial products: fp32's exponent range, 24-bit significands - fp32 semantics per element.
#      The library default: a direct caller of the operators gets this.
#   2  two fp16 planes of the operands scaled to their largest finite magnitude, three partial products: 22-bit
#      significands, absolute accuracy 2^-40 of a tensor's largest magnitude per element - half the matrix work. What
#      ``train.Runner`` selects for the train step, under the ``RangeGuard`` below that measures the step's real operands
#      and goes back to 3 when they are not the kind of tensor the two-plane form represents as well as fp32 does.
# GGA_DENSE_PLANES in the environment fixes the choice (the Runner then leaves it alone).
PLANES = int(os.environ.get('GGA_DENSE_PLANES', '3'))
PLANES_PINNED = 'GGA_DENSE_PLANES' in os.environ
FELL_BACK = False       # a range guard of this process has left the two-plane form (train.Runner): later Runners start on three
BN_BWD_FUSED = True     # backward-data convolutions reduce the BatchNorm backward sums of the layer below them (BnSource)


class RangeGuard:
    """Measures, for every operand the matrix kernels are handed while it is armed, how much of the tensor lies where
    the two-plane form is worse than fp32: with the scale taken from the largest magnitude ``amax`` an element keeps an
    absolute accuracy of 2^-40 * amax, so a non-zero element below 2^-17 * amax has fewer than fp32's 24 significant
    bits and one below 2^-30 * amax is off by more than 1e-3 of itself (``lost``). Recorded per operand, on the device,
    without a host read: [non-zero count, count below 2^-17 amax, count below 2^-30 amax, sum |t|, sum of |t| below
    2^-30 amax, amax]. ``Runner`` arms it for one step every ``interval`` iterations, reads the rows once after that step
    and leaves the two-plane form when an operand is over its limit:

    * operands of the FORWARD pass (activations, weights): more than ``LIMIT`` of the non-zero ELEMENTS lost. What follows
      a forward convolution is non-linear - a normalisation layer rescales a sample or channel whose values are all tiny to
      O(1) - so every element has to be right by itself;
    * operands of the BACKWARD pass (gradients): more than ``LIMIT_MASS`` of the tensor's L1 MASS lost. Everything
      downstream of a gradient is linear in it (backward-data, BatchNorm / GroupNorm backward, the weight-gradient sums),
      so the error any later value inherits is bounded by that share; a gradient tensor routinely has whole regions 2^-30
      below its largest entry (locations far from every object) that decide nothing."""
    LIMIT = 1e-3
    LIMIT_MASS = 1e-6

    def __init__(self):
        self.armed = False
        self.phase = 'forward'
        self.rows, self.names = [], []

    def arm(self):
        self.armed, self.phase, self.rows, self.names = True, 'forward', [], []

    def record(self, t, amax=None):
        if not self.armed or t.numel() == 0:
            return
        with torch.no_grad():
            # reductions without full-size float64 copies or stacked boolean temporaries: at most two fp32 temporaries of the
            # operand's size are alive at a time (the 16 x 384 x 248 x 216 concat is 1.3 GB), sums accumulate in float64
            a = t.detach().abs().reshape(-1)
            a = torch.where(torch.isfinite(a), a, a.new_zeros(()))
            m = a.max()
            f64 = dict(dtype=torch.float64)
            nz = (a > 0).sum(**f64)
            low17 = ((a > 0) & (a < m * 2.0 ** -17)).sum(**f64)
            lost = torch.where(a < m * 2.0 ** -30, a, a.new_zeros(()))          # zeros stay zero: count and mass from one temporary
            self.rows.append(torch.stack([nz, low17, (lost > 0).sum(**f64), a.sum(**f64), lost.sum(**f64), m.double()]))
            del a, lost
            self.names.append((f'{tuple(t.shape)}', self.phase))

    def disarm(self):
        """-> list of dicts (one per operand seen): shape, phase, nonzero, share_below_2^-17, share_lost, mass_lost, amax,
        over (the operand is over the limit of its phase)."""
        self.armed = False
        if not self.rows:
            return []
        r = torch.stack(self.rows).cpu()        # the one host read
        out = []
        for (name, phase), (nz, l17, l30, s, s30, m) in zip(self.names, r.tolist()):
            d = dict(shape=name, phase=phase, nonzero=int(nz), share_below_2p17=l17 / max(nz, 1), share_lost=l30 / max(nz, 1),
                     mass_lost=s30 / s if s > 0 else 0.0, amax=m)
            d['over'] = d['share_lost'] > self.LIMIT if phase == 'forward' else d['mass_lost'] > self.LIMIT_MASS
            out.append(d)
        self.rows, self.names = [], []
        return out


RANGE_GUARD = RangeGuard()


def _amax_bits(t):
    L = _lib.lib()
    out = torch.empty(1, dtype=torch.int32, device=t.device)
    if not (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
        t = t.contiguous()
    n = t.numel()
    if n % 4 or t.data_ptr() % 16:           # odd sizes (e.g. 5-feature sparse inputs): the framework's reduction
        a = t.detach().abs()
        m = torch.where(torch.isfinite(a), a, torch.zeros_like(a)).max() if n else t.new_zeros(())
        return m.reshape(1).view(torch.int32)
    width = 1024 if n % 1024 == 0 and n else max(n, 4)
    assert width < 2 ** 31
    check(L.gga_absmax_bits(F._p(t), n // width, width, width, F._p(out), F._stream()), 'gga_absmax_bits')
    return out


def amax_bits(t):
    """Device scalar (int32 tensor [1]) with the bits of the largest finite ``|t|`` - what the two-plane kernels
    derive their power-of-two scale from. ``t``: any tensor that is dense in memory, or a row-major matrix view."""
    RANGE_GUARD.record(t)
    return _amax_bits(t)


def _cdiv(a, b):
    return -(-a // b)


def _transposed(H, W):
    """Walk the map with the 32-pixel tile edge along H instead of W when that wastes less of the
    8 x 32 tiles (124 x 108: 1.07 instead of 1.22 tiles of work per tile of data)."""
    normal = _cdiv(W, 32) * 32 * _cdiv(H, 8) * 8
    turned = _cdiv(H, 32) * 32 * _cdiv(W, 8) * 8
    return turned < 0.97 * normal


WS_LEVELS = os.environ.get('GGA_DC_WS_LEVELS', '1') != '0'      # A/B switch: multi-map launches of the lock-step kernel instead


class _AmaxPool:
    """Zeroed int32 slots for the producers' absmax outputs: one memset per generation instead of one fill launch per
    BatchNorm call (57 per PointPillars step). ``next_generation()`` (Runner.step) zeroes the pool again; a tensor's
    remembered absmax is only valid in the generation it was produced in."""
    SLOTS = 1024

    def __init__(self):
        self.buf, self.used, self.generation = {}, {}, 0

    def next_generation(self):
        self.generation += 1
        for dev, b in self.buf.items():
            if self.used.get(dev, 0):
                b.zero_()
                self.used[dev] = 0

    def take(self, device, n=1):
        """``n`` consecutive zeroed slots (one absmax per 64-channel block of a tensor, see ``_wgrad``)."""
        key = str(device)
        b = self.buf.get(key)
        if b is None:
            b = self.buf[key] = torch.zeros(self.SLOTS, dtype=torch.int32, device=device)
            self.used[key] = 0
        i = self.used[key]
        if i + n > self.SLOTS:                    # more producers than slots in one generation: a fill of its own
            return torch.zeros(n, dtype=torch.int32, device=device)
        self.used[key] = i + n
        return b[i:i + n]


AMAX_POOL = _AmaxPool()


def set_amax(t, amax):
    """Remember the absmax bits a producer kernel left for ``t`` (valid while ``t`` is not written again and the pool
    slot has not been recycled)."""
    if amax is not None:
        t._gga_amax = (t._version, t.data_ptr(), t.numel(), amax, AMAX_POOL.generation)
    return t


def new_amax(device, n=1):
    """Zeroed absmax accumulator(s) for the producers' ``amax`` outputs (None on the three-bf16-plane path)."""
    return AMAX_POOL.take(device, n) if PLANES == 2 else None


def tensor_amax(t):
    """Absmax bits of ``t``: what its producer left (``set_amax``) if ``t`` has not been modified since, one
    ``gga_absmax_bits`` pass otherwise."""
    RANGE_GUARD.record(t)
    c = getattr(t, '_gga_amax', None)
    if (c is not None and c[0] == t._version and c[1] == t.data_ptr() and c[2] == t.numel()
            and c[4] == AMAX_POOL.generation):
        return c[3]
    return _amax_bits(t)


class BnSource:
    """What a backward-data convolution needs to take the reduce pass of a BatchNorm backward into its epilogue
    (``gga_dense_conv3x3_bn_bwd``): ``z = relu(bn(x))`` with ``parts`` = [(first channel of z, channels, x, gamma, beta,
    saved mean / invstd)] - one entry, or one per concatenated branch. Attached to ``z`` by ``functional.bn_act`` /
    ``bn_relu_cat`` and valid while ``z`` has not been written again."""

    def __init__(self, z, parts):
        self.key = (z._version, z.data_ptr(), z.numel())
        self.parts = parts

    def valid_for(self, z):
        return self.key == (z._version, z.data_ptr(), z.numel())

    def part(self, c0, width):
        """(x pointer, x pixel stride, gamma, beta, mean, invstd pointers) of channels [c0, c0 + width), or None when
        they straddle two branches."""
        for start, C, x, gamma, beta, saved in self.parts:
            if start <= c0 and c0 + width <= start + C:
                o = 4 * (c0 - start)
                return (x.data_ptr() + o, C, gamma.data_ptr() + o, beta.data_ptr() + o, saved.data_ptr() + o,
                        saved.data_ptr() + 4 * C + o)
        return None

    def covers(self, n_out):
        if len(self.parts) == 1 and self.parts[0][1] == n_out and n_out <= 128:
            return True
        widths = [n_out] if n_out in (64, 128) else [128] * (n_out // 128)
        return n_out == sum(C for _, C, *_ in self.parts) and all(
            self.part(c0, w) is not None for c0, w in zip(range(0, n_out, 128), widths))


def bn_source(z, n_out):
    """The valid ``BnSource`` of ``z`` for a backward-data convolution with ``n_out`` output channels, or None (also
    where the epilogue would cost more than the reduce pass it replaces: ``gga_dense_conv3x3_bn_bwd_pays_planes``). ``z``
    [rows, C] (sparse features): the gather-GEMM kernel's epilogue, one launch of at most 128 channels."""
    src = getattr(z, '_gga_bn_src', None)
    if src is None or not src.valid_for(z) or not src.covers(n_out):
        return None
    if z.dim() == 2:
        return src if (BN_BWD_FUSED and n_out <= 128 and len(src.parts) == 1) else None
    B, _, H, W = z.shape
    th, tw = (W, H) if _transposed(H, W) else (H, W)
    if not _lib.lib().gga_dense_conv3x3_bn_bwd_pays_planes(B, th, tw, 64 if n_out == 64 else 128, PLANES):
        return None
    return src


class BnPartials:
    """Left on a gradient by the backward-data convolution that already reduced it (see BnSource): the gradient is
    masked by the ReLU and ``parts`` = [(first channel, channels, [tiles, 2, channels] f64 sums)]."""

    def __init__(self, g, parts, tokens):
        self.key = (g._version, g.data_ptr(), g.numel())
        self.parts, self.tokens = parts, tokens

    def take(self, g, token, c0, C):
        """The partial sums of channels [c0, c0 + C) if ``g`` is still the tensor the convolution wrote and ``token``
        (data pointer of the BatchNorm's saved statistics) is the one it masked with."""
        if self.key != (g._version, g.data_ptr(), g.numel()) or token not in self.tokens:
            return None
        hit = [p for s, w, p in self.parts if c0 <= s and s + w <= c0 + C]
        if sum(p.shape[2] for p in hit) != C:
            return None
        return hit[0] if len(hit) == 1 else torch.cat(hit, dim=2)


def _operand(weight, backward, transposed, planes, c0=None):
    """(split-plane operand, absmax slot or None) of the forward (or backward-data) convolution from the weight bank: the
    operands persist and are refreshed together once per optimizer step (``weight_bank``). ``weight``: the parameter
    [cout, cin, 3, 3] in any memory layout, or a LIST of such whose concatenation along the output channels is meant (never
    materialised). ``c0``: the 128-channel slice [c0, c0 + 128) of the convolution's output."""
    from . import weight_bank
    if isinstance(weight, (list, tuple)):
        assert c0 is None
        return weight_bank.dense_operand_cat(weight, backward, transposed, planes)
    return weight_bank.dense_operand(weight, backward, transposed, planes, c0)


def _run(x, weight, backward, want_stats=False, x_amax=None, w_amax=None, y=None, y_col=0, bn=None):
    """The convolution (or its backward-data form) of ``x`` with ``weight`` [cout, cin, 3, 3] (or a list of weights standing
    for their concatenation along cout); output widths above 128 run as 128-channel slices of the result. ``x_amax``: absmax
    bits of ``x`` when already known (two-plane arithmetic; computed here otherwise); the weight's operand and absmax come
    from the weight bank (``w_amax`` is ignored). ``y`` / ``y_col``: write the result into the channel block starting at
    ``y_col`` of an existing channels-last tensor. ``bn`` (a ``BnSource``, backward only): the result is the gradient of that
    BatchNorm + ReLU output - it is stored masked by the ReLU, and the second return value is then the list [(first channel,
    channels, per-tile sums of g and g * xhat)] (``BnPartials.parts``)."""
    B, n_in, H, W = x.shape
    L = _lib.lib()
    many = isinstance(weight, (list, tuple))
    cout_all = sum(w.shape[0] for w in weight) if many else weight.shape[0]
    cin_all = weight[0].shape[1] if many else weight.shape[1]
    n_out = cin_all if backward else cout_all
    tr = _transposed(H, W)
    if y is None:
        y = torch.empty((B, n_out, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    ystride = y.shape[1]
    planes = PLANES
    if planes == 2:
        x_amax = amax_bits(x) if x_amax is None else x_amax
        if RANGE_GUARD.armed:
            for w in (weight if many else [weight]):
                RANGE_GUARD.record(w.detach())
    else:
        x_amax = None
    stats = None
    want_stats = want_stats or bn is not None
    none6 = (None, 0, None, None, None, None)
    if n_out in (64, 128):
        if want_stats:
            tiles = int(L.gga_dense_conv3x3_stat_rows(B, W, H, n_out, planes, 1) if tr else L.gga_dense_conv3x3_stat_rows(B, H, W, n_out, planes, 1))
            stats = torch.empty((tiles, 2, n_out), dtype=torch.float64, device=x.device)
        wp, wa = _operand(weight, backward, tr, planes)
        check(L.gga_dense_conv3x3_bn_bwd(F._p(x), F._p(wp), B, H, W, n_in, n_out,
                                         y.data_ptr() + 4 * y_col, ystride, int(tr), F._p(stats), planes, F._p(x_amax),
                                         F._p(wa), *(bn.part(0, n_out) if bn else none6), F._stream()), 'gga_dense_conv3x3')
        if bn:
            stats = [(0, n_out, stats)]
    elif bn is None and n_out // 128 <= 16:
        # every 128-channel slice of the output in ONE launch (gga_dense_conv3x3_levels with the slices as entries): a
        # 62 x 54 map is 224 tiles per slice - half of what the chip holds at once
        import ctypes as C
        assert not many
        n = n_out // 128
        th, tw = (W, H) if tr else (H, W)
        rows = int(L.gga_dense_conv3x3_tile_rows(B, th, tw, 128, planes))      # two planes: 8 (the producer / consumer form takes the slices as one grid)
        tiles = int(L.gga_dense_conv3x3_stat_rows(B, th, tw, 128, planes, n))       # rows of each slice's statistics
        ops = [_operand(weight, backward, tr, planes, c0) for c0 in range(0, n_out, 128)]      # slices of one weight: one absmax slot
        wa = ops[0][1]
        sts = [torch.empty((tiles, 2, 128), dtype=torch.float64, device=x.device) for _ in range(n)] if want_stats else None
        vp, i32 = C.c_void_p * n, C.c_int32 * n
        check(L.gga_dense_conv3x3_levels(
            n, vp(*[x.data_ptr()] * n), i32(*[H] * n), i32(*[W] * n), vp(*[w.data_ptr() for w, _ in ops]), B, n_in, 128,
            vp(*[y.data_ptr() + 4 * (y_col + c0) for c0 in range(0, n_out, 128)]), ystride, planes,
            vp(*[x_amax.data_ptr()] * n) if planes == 2 else None, F._p(wa), None, rows, int(tr),
            vp(*[t.data_ptr() for t in sts]) if sts else None, F._stream()), 'gga_dense_conv3x3_levels')
        if sts:
            stats = torch.cat(sts, dim=2)           # [tiles, 2, n_out]
    else:
        assert not many
        parts = []
        for c0 in range(0, n_out, 128):             # strided views: packed straight from the parameter
            wp, wa = _operand(weight, backward, tr, planes, c0)
            st = None
            if want_stats:                          # per-channel sums of this 128-channel block of the output
                tiles = int(L.gga_dense_conv3x3_stat_rows(B, W, H, 128, planes, 1) if tr else L.gga_dense_conv3x3_stat_rows(B, H, W, 128, planes, 1))
                st = torch.empty((tiles, 2, 128), dtype=torch.float64, device=x.device)
                parts.append((c0, 128, st))
            check(L.gga_dense_conv3x3_bn_bwd(F._p(x), F._p(wp), B, H, W, n_in, 128,
                                             y.data_ptr() + 4 * (y_col + c0), ystride, int(tr), F._p(st), planes, F._p(x_amax),
                                             F._p(wa), *(bn.part(c0, 128) if bn else none6), F._stream()),
                  'gga_dense_conv3x3_slice')
        if bn:
            stats = parts
        elif parts:
            stats = torch.cat([p for _, _, p in parts], dim=2)         # [tiles, 2, n_out]
    return y, stats


def run_bn_bwd(gy, weight, g_amax, w_amax, src):
    """Backward-data convolution of ``gy`` whose result is the gradient of the BatchNorm + ReLU output described by
    ``src`` (a ``BnSource`` or None): with a source the result carries the ``BnPartials`` its BatchNorm backward takes."""
    gx, parts = _run(gy, weight, True, False, g_amax, w_amax, bn=src)
    if src is not None:
        gx._gga_bn_bwd = BnPartials(gx, parts, tuple(p[5].data_ptr() for p in src.parts))
    return gx


def _wgrad(x, gy, weight, x_amax=None, g_amax=None, g_per_block=False):
    """Weight gradient on the matrix path (``gga_dense_wgrad3x3``), in the parameter's memory layout. ``g_per_block``:
    ``g_amax`` holds one absmax per 64-channel block of ``gy`` (two-plane arithmetic)."""
    L = _lib.lib()
    B, cin, H, W = x.shape
    if isinstance(weight, (list, tuple)):        # the concatenation of these weights along cout: one gradient tensor for all
        cout = sum(w.shape[0] for w in weight)
        gw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device).contiguous(memory_format=torch.channels_last)
    else:
        cout = weight.shape[0]
        gw = torch.empty_like(weight)
    s = gw.stride()
    ws = F._workspace('dense_wgrad', L.gga_dense_wgrad3x3_workspace_bytes(B, H, W, cin, cout), x.device)
    tr = _cdiv(H, 32) * 32 * W < 0.97 * _cdiv(W, 32) * 32 * H        # 32-pixel strips along H waste less
    planes = PLANES
    if planes == 2:
        x_amax = amax_bits(x) if x_amax is None else x_amax
        g_amax = amax_bits(gy) if g_amax is None else g_amax
    else:
        x_amax = g_amax = None
    check(L.gga_dense_wgrad3x3_block_amax(F._p(x), F._p(gy), B, H, W, cin, cout, F._p(gw), s[0], s[1], s[2], s[3], int(tr), planes,
                                          F._p(x_amax), 0, F._p(g_amax), int(bool(g_per_block) and planes == 2), F._p(ws), ws.numel(),
                                          F._stream()), 'gga_dense_wgrad3x3')
    return gw


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, want_stats):
        cout, cin = weight.shape[0], weight.shape[1]
        two = PLANES == 2
        x_amax = tensor_amax(x) if two else None
        y, stats = _run(x, weight.detach(), False, want_stats, x_amax)
        ctx.save_for_backward(x, weight)
        ctx.amax = (x_amax, None)
        # x = relu(bn(.)) with this convolution as its consumer: the backward-data pass then does that BatchNorm's reduce
        ctx.bn_src = bn_source(x, cin) if BN_BWD_FUSED else None
        if stats is None:
            stats = torch.empty(0, dtype=torch.float64, device=x.device)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)     # no zero tensors (one fill launch each) for the gradients of the non-differentiable outputs
        return y, stats

    @staticmethod
    def backward(ctx, gy, _gstats):
        x, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = None
        x_amax, w_amax = ctx.amax
        g_amax = tensor_amax(gy) if PLANES == 2 else None        # left by gy's producer, or one pass for both consumers
        if PLANES != 2:
            x_amax = w_amax = None
        # backward-data is a cout -> cin convolution (taps reversed, channel roles swapped): cin is its
        # output width; wider inputs are produced in 128-channel slices
        mine = cin in (64, 128) or cin % 128 == 0
        if ctx.needs_input_grad[0] and mine:
            gx = run_bn_bwd(gy, weight.detach(), g_amax, w_amax, ctx.bn_src)
        need_gx = ctx.needs_input_grad[0] and not mine
        need_gw = bool(ctx.needs_input_grad[1])
        if need_gw and WGRAD and cin % 64 == 0 and cout % 64 == 0:
            gw = _wgrad(x, gy, weight, x_amax, g_amax)
            need_gw = False
        if need_gw or need_gx:
            r = torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                    [need_gx, need_gw, False])
            if need_gw:
                gw = r[1]
            if need_gx:
                gx = r[0]
        return gx, gw, None


def _run_levels(xs, weight, backward, x_amaxes, w_amax, bias=None):
    """The convolution (or its backward-data form) of several maps ``xs`` [B, n_in, H_l, W_l] with ONE ``weight`` (and
    ``bias``, added in the kernel's epilogue): one launch over all (map, 128-channel output slice) entries
    (``gga_dense_conv3x3_levels``: at most 16 entries per launch)."""
    import ctypes as C
    L = _lib.lib()
    B, n_in = xs[0].shape[0], xs[0].shape[1]
    n_out = weight.shape[1] if backward else weight.shape[0]
    width = n_out if n_out in (64, 128) else 128
    ys = [torch.empty((B, n_out, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
          for x in xs]
    planes = PLANES
    packs, w_amax = [], None
    for c0 in range(0, n_out, width):
        wp, w_amax = _operand(weight, backward, False, planes, None if width == n_out else c0)
        packs.append((c0, wp))
    if planes == 2 and RANGE_GUARD.armed:
        RANGE_GUARD.record(weight)
    entries = [(x, y, c0, wp, a) for c0, wp in packs for x, y, a in zip(xs, ys, x_amaxes)]
    # maps with enough 16-row tiles run the 512-thread form (as gga_dense_conv3x3_planes would pick for them), the small
    # ones the 8-row form: two launches
    big = lambda x: width == 128 and B * _cdiv(x.shape[3], 32) * _cdiv(x.shape[2], 16) >= 384
    groups = [(16, [e for e in entries if big(e[0])]), (8, [e for e in entries if not big(e[0])])]
    if (WS_LEVELS and planes == 2 and width == 128 and bias is None
            and L.gga_dense_conv3x3_tile_rows(B, xs[0].shape[2], xs[0].shape[3], 128, 2) == 8):
        # the producer / consumer form (two planes, GGA_DC_WS != 0) takes the 128-channel slices of ONE map as one grid: a launch per map
        groups = [(8, [e for e in entries if e[0] is x]) for x in xs]
    for rows, group in groups:
        for i0 in range(0, len(group), 16):
            part = group[i0:i0 + 16]
            n = len(part)
            vp, i32 = C.c_void_p * n, C.c_int32 * n
            amax = vp(*[a.data_ptr() for *_, a in part]) if planes == 2 else None
            bias_p = vp(*[bias.data_ptr() + 4 * c0 for _, _, c0, _, _ in part]) if bias is not None else None
            check(L.gga_dense_conv3x3_levels(
                n, vp(*[x.data_ptr() for x, *_ in part]), i32(*[x.shape[2] for x, *_ in part]), i32(*[x.shape[3] for x, *_ in part]),
                vp(*[wp.data_ptr() for _, _, _, wp, _ in part]), B, n_in, width,
                vp(*[y.data_ptr() + 4 * c0 for _, y, c0, _, _ in part]), n_out, planes, amax,
                F._p(w_amax) if planes == 2 else None, bias_p, rows, 0, None, F._stream()), 'gga_dense_conv3x3_levels')
    return ys


class _Conv3x3Levels(torch.autograd.Function):
    """One 3x3 convolution applied to several maps (the FPN levels of a head tower): forward and backward-data are one
    launch over all levels, the weight gradient is the sum of the levels' weight gradients."""

    @staticmethod
    def forward(ctx, weight, bias, *xs):
        two = PLANES == 2
        x_amaxes = [tensor_amax(x) if two else None for x in xs]
        ys = _run_levels(xs, weight.detach(), False, x_amaxes, None, None if bias is None else bias.detach().contiguous())
        ctx.save_for_backward(weight, *xs)
        ctx.amax = (x_amaxes, None)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        weight, *xs = ctx.saved_tensors
        two = PLANES == 2
        gys = [g.contiguous(memory_format=torch.channels_last) for g in gys]
        x_amaxes, w_amax = ctx.amax
        g_amaxes = [tensor_amax(g) if two else None for g in gys]
        gxs = [None] * len(xs)
        if any(ctx.needs_input_grad[2:]):
            gxs = _run_levels(gys, weight.detach(), True, g_amaxes, w_amax)
        gw = gb = None
        if ctx.needs_input_grad[0]:
            for x, g, xa, ga in zip(xs, gys, x_amaxes, g_amaxes):
                part = _wgrad(x, g, weight, xa, ga)
                gw = part if gw is None else gw.add_(part)
        if ctx.needs_input_grad[1]:
            for g in gys:
                part = F.channel_sums(g)
                gb = part if gb is None else gb.add_(part)
        return (gw, gb, *gxs)


def levels_eligible(conv, xs):
    """One multi-level launch is possible: an eligible convolution (with or without bias) over maps of equal batch whose
    backward-data and weight gradient also run on the matrix kernels."""
    return (len(xs) >= 1 and all(eligible(conv, x) for x in xs) and WGRAD
            and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0
            and (conv.in_channels in (64, 128) or conv.in_channels % 128 == 0)
            and all(x.shape[0] == xs[0].shape[0] for x in xs))


def conv2d_levels(xs, conv):
    """``[conv(x) for x in xs]`` - one launch over all maps when ``levels_eligible``, map by map otherwise."""
    if levels_eligible(conv, xs):
        return list(_Conv3x3Levels.apply(conv.weight, conv.bias, *xs))
    return [conv2d(x, conv) for x in xs]


def eligible(conv, x):
    return (ENABLED and type(conv) is nn.Conv2d and conv.kernel_size == (3, 3)
            and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding_mode == 'zeros' and conv.in_channels % 32 == 0
            and (conv.out_channels in (64, 128) or conv.out_channels % 128 == 0)
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and x.shape[2] * x.shape[3] * x.shape[1] < 2 ** 31)


def conv2d(x, conv, bn_follows=False):
    """``conv(x)``: the bf16x9 kernel for the 64/128-channel 3x3 convolutions, the module otherwise.
    ``bn_follows``: the caller normalises the result with training-mode batch statistics next; the
    kernel then also leaves the per-channel sums of its output (``y.bn_partials``), which
    ``functional.bn_act`` / ``bn_relu_head_conv3x3`` use instead of re-reading ``y``."""
    if eligible(conv, x):
        if conv.bias is not None:         # bias in the kernel's epilogue (gga_dense_conv3x3_levels with one map: mono3d heads)
            if levels_eligible(conv, [x]):
                return _Conv3x3Levels.apply(conv.weight, conv.bias, x)[0]
            return _Conv3x3.apply(x, conv.weight, False)[0] + conv.bias.view(1, -1, 1, 1)
        y, stats = _Conv3x3.apply(x, conv.weight, bool(bn_follows))
        if bn_follows:
            F.attach_bn_partials(y, stats)
        return y
    from . import strided_conv
    if strided_conv.eligible(conv, x):        # stride-2 3x3, 1x1 and kernel = stride transposed convolutions
        return strided_conv.conv(x, conv)
    return conv(x)
