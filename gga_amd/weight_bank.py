"""Packed operands of the split-plane matrix kernels for all convolution weights, refreshed by two launches per step.

Every convolution on the matrix kernels needs its weight as a packed operand (16-bit planes in the kernel's stage order, one
operand per direction - forward / backward-data -, per tile walk and per 128-column slice) and, on two fp16 planes, the weight's
absmax. Rounds 1-3 made them at every use: one ``absmax_kernel`` and one ``*_pack_weight_kernel`` per convolution and direction
and step (79 launches of ~5 us in the PointPillars step, ~150 in the sparse step, plus the ``torch.cat`` / ``permute().reshape()``
copies that fed them). Weights change once per step - in the optimizer - and all of them together, so here the operands persist:
the first request after a weight changed (its version counter) refreshes EVERY stale operand of the bank with one
``gga_absmax_table`` and one ``gga_pack_weights_table`` launch over device-side tables (csrc/weight_bank.hip), and the rest of
the step's requests are dictionary lookups.

An operand is described by its sources: 4-D strided VIEWS ``[k0, k1, channel, column]`` of the parameters themselves (a
``permute`` of a Conv2d weight - never a copy), each with the channel / column offset where it lands, so concatenations are
virtual: the two first convolutions a head-branch launch covers, the 15 branch weights of the 960 -> 64 backward-data
convolution. Sources that form one operand share one absmax slot (the scale of the operand).

Streams: the refresh runs on the stream of the request that triggered it; a request from another stream waits for the refresh's
event. Operands nobody asked for during the last ``KEEP`` refreshes are dropped (models come and go in a test process); in a
process whose weights never change (inference, pseudo-label runs) the same sweep runs every ``SWEEP_EVERY`` operand creations.

What tells the bank that weights changed: the tensors' version counters (in-place torch ops, ``load_state_dict``, non-fused
optimizers) and ``invalidate()``, which a process-wide optimizer-step hook (registered below, at import) calls after the step of
EVERY ``torch.optim`` optimizer - the fused ones leave the version counters alone. Updates that do neither - writes through
``param.data`` (``p.data.copy_``, EMA / SWA through ``.data``), raw-pointer kernels - need an explicit ``BANK.invalidate()``.
``GGA_BANK_VERIFY=1`` checks every request against the sources as they are now (a synchronising debug switch) and raises on a
stale operand."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from . import functional as F
from ._lib import check

LAYOUT_GATHER, LAYOUT_DENSE = 0, 1
ENABLED = True          # False: every request packs on the spot (one single-entry table per request; the A/B switch)
VERIFY = os.environ.get('GGA_BANK_VERIFY') == '1'


class PackEntry(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('amax', C.c_void_p), ('s_k0', C.c_int64), ('s_k1', C.c_int64),
                ('s_c', C.c_int64), ('s_col', C.c_int64), ('first', C.c_int64), ('kw', C.c_int32), ('kvol', C.c_int32),
                ('n_c', C.c_int32), ('n_col', C.c_int32), ('c0', C.c_int32), ('col0', C.c_int32), ('n_in', C.c_int32),
                ('co', C.c_int32), ('layout', C.c_int32), ('reverse', C.c_int32)]


class AmaxEntry(C.Structure):
    _fields_ = [('src', C.c_void_p), ('slot', C.c_void_p), ('n', C.c_int64), ('first_block', C.c_int64)]


def _memory_block(t):
    """(address, floats) of the block of memory ``t``'s elements occupy when they are dense in it (any permutation of a
    contiguous tensor: a channels-last parameter, a slice along its slowest axis); otherwise the tensor's whole storage."""
    dims = sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1)
    run = 1
    for st, sz in dims:
        if st != run:
            store = t.untyped_storage()
            return (store.data_ptr(), store.nbytes() // 4)
        run *= sz
    return (t.data_ptr(), t.numel())


def _padded_columns(n_out):
    return 32 * (1 if n_out <= 32 else (2 if n_out <= 64 else 4))


class _Operand:
    __slots__ = ('key', 'sources', 'n_in', 'n_out', 'layout', 'reverse', 'planes', 'packed', 'slot', 'versions', 'used', 'blocks', 'epoch',
                 'touched', 'digest')

    def stale(self, epoch):
        return self.epoch != epoch or any(v._version != ver for (v, _, _), ver in zip(self.sources, self.versions))


class WeightBank:
    KEEP = 4
    SLOTS = 2048
    SWEEP_EVERY = 256           # operand creations between two sweeps of a process whose weights never change

    def __init__(self):
        self.ops = {}
        self.slots = {}             # device -> int32 [SLOTS]
        self.slot_of = {}           # (device, memory blocks an operand's scale is taken over) -> slot index
        self.slot_users = {}        # (device, slot index) -> operands holding it (a slot returns to `free` with its last user)
        self.free = {}              # device -> released slot indices
        self.slot_token = {}        # (device, slot index) -> (epoch, versions) its content was computed for
        self.refreshes = 0          # table launches (diagnostics)
        self.generation = 0         # refreshes triggered by changed weights: one per optimizer step in a train loop
        self.tables = {}            # device -> (signature of the stale set, pack table, amax table, totals)
        self.event, self.stream = None, None
        self.epoch = 0              # bumped by invalidate(): every operand made before is stale
        self.created = 0            # operands ever made; `created // SWEEP_EVERY` is the sweep period an operand was last touched in

    def invalidate(self):
        """Every weight may have changed. Needed after updates that do not bump the tensors' version counters - torch's FUSED
        optimizers do not (``torch.optim.AdamW(fused=True)`` leaves ``param._version`` alone) -; ``train.build_optimizer``
        registers it as a step hook of every optimizer it builds. In-place torch ops, ``load_state_dict`` and non-fused
        optimizers are noticed through the version counters without it."""
        self.epoch += 1

    # ------------------------------------------------------------------------------------------------------------ requests
    def operand(self, sources, n_in, n_out, layout, reverse, planes, scale_of=None):
        """``sources``: [(view [k0, k1, n_c, n_col], c0, col0), ...] -> (packed int16 tensor, absmax slot int32 [1] or None).
        ``scale_of``: the tensors whose largest magnitude is the operand's scale (default: the sources themselves) - the whole
        weight when the sources are slices of it, so that all slices of a convolution share one scale."""
        assert n_out <= 128
        blocks = tuple(sorted({_memory_block(t) for t in (scale_of if scale_of is not None else [v for v, _, _ in sources])})) if planes == 2 else ()
        key = (planes, layout, bool(reverse), n_in, n_out, blocks,
               tuple((v.data_ptr(), tuple(v.shape), tuple(v.stride()), c0, col0) for v, c0, col0 in sources))
        op = self.ops.get(key)
        if op is None:
            op = self._new(key, sources, n_in, n_out, layout, reverse, planes, blocks)
            self._refresh([op])
        elif op.stale(self.epoch):
            stale = [o for o in self.ops.values() if o.stale(self.epoch)] if ENABLED else [op]
            self._refresh(stale)
            self.generation += 1
            if self.generation % 8 == 0:          # operands nobody asked for in a while (their model is gone)
                dead = [k for k, o in self.ops.items() if o.used < self.generation - self.KEEP * 8]
                for k in dead:
                    self._release(self.ops.pop(k))
                if dead:
                    self.tables.clear()
        op.used = self.generation
        op.touched = self.created // self.SWEEP_EVERY
        if VERIFY:
            self._verify(op)
        if self.event is not None:
            cur = torch.cuda.current_stream(op.packed.device)
            if cur.cuda_stream != self.stream:
                cur.wait_event(self.event)
        return op.packed, op.slot

    def _new(self, key, sources, n_in, n_out, layout, reverse, planes, blocks):
        L = _lib.lib()
        dev = sources[0][0].device
        op = _Operand()
        op.key, op.sources, op.n_in, op.n_out, op.layout, op.reverse, op.planes = key, list(sources), n_in, n_out, layout, bool(reverse), planes
        kvol = sources[0][0].shape[0] * sources[0][0].shape[1]
        op.packed = torch.zeros(L.gga_sparse_split_weight_bytes(kvol, n_in, n_out) // 2, dtype=torch.int16, device=dev)
        op.versions = [-1] * len(sources)
        op.epoch = -1
        op.used = self.generation
        op.digest = None
        self.created += 1
        op.touched = period = self.created // self.SWEEP_EVERY
        if self.created % self.SWEEP_EVERY == 0:
            # weights that never change never reach the sweep of `operand`: drop what nobody asked for during the whole
            # previous period (a live model asks for its operands in every pass; one dropped by mistake is simply made again)
            dead = [k for k, o in self.ops.items() if o.touched < period - 1]
            for k in dead:
                self._release(self.ops.pop(k))
            if dead:
                self.tables.clear()
        op.slot, op.blocks = None, []
        if planes == 2:
            d = str(dev)
            skey = (d, tuple(blocks))
            if d not in self.slots:
                self.slots[d] = torch.zeros(self.SLOTS, dtype=torch.int32, device=dev)
                self.free[d] = list(range(self.SLOTS - 1, -1, -1))
            if skey not in self.slot_of:
                if not self.free[d]:
                    raise RuntimeError(f'weight bank: all {self.SLOTS} absmax slots of {d} are held by live operands '
                                       f'({len(self.ops)} operands) - raise WeightBank.SLOTS')
                self.slot_of[skey] = self.free[d].pop()
            i = self.slot_of[skey]
            self.slot_users[(d, i)] = self.slot_users.get((d, i), 0) + 1
            op.slot, op.blocks = self.slots[d][i:i + 1], [(p, n, i) for p, n in blocks]
        self.ops[key] = op
        return op

    def _release(self, op):
        if op.slot is None:
            return
        d, i = str(op.packed.device), op.blocks[0][2]
        self.slot_users[(d, i)] -= 1
        if self.slot_users[(d, i)] == 0:
            del self.slot_users[(d, i)]
            self.slot_token.pop((d, i), None)
            self.slot_of = {k: v for k, v in self.slot_of.items() if not (k[0] == d and v == i)}
            self.free[d].append(i)

    # ------------------------------------------------------------------------------------------------------------- refresh
    def _refresh(self, stale):
        L = _lib.lib()
        by_dev = {}
        for op in stale:
            by_dev.setdefault((str(op.packed.device), op.planes), []).append(op)
        for (dev_key, planes), ops in by_dev.items():
            dev = ops[0].packed.device
            # absmax slots to recompute: those whose content was not computed for the weights as they are now. (A slot is shared
            # by every operand scaled by the same weight - forward / backward-data, slices; an operand made later in the same
            # step must not zero and recompute a slot that kernels of another stream may be reading.)
            tokens = {(dev_key, b[2]): (self.epoch, tuple(v._version for v, _, _ in op.sources)) for op in ops for b in op.blocks[:1]}
            need = frozenset(i for (d, i), tok in tokens.items() if self.slot_token.get((d, i)) != tok)
            sig = (planes, tuple(id(o) for o in ops), need)
            cached = self.tables.get((dev_key, planes))
            if cached is None or cached[0] != sig:
                cached = (sig,) + self._build_tables(ops, dev, planes, need)
                if len(ops) > 1:
                    self.tables[(dev_key, planes)] = cached
            self.slot_token.update(tokens)
            _, pack_t, n_pack, total, amax_t, n_amax, n_blocks, slot_idx = cached
            with torch.cuda.device(dev):
                if planes == 2 and n_amax:
                    # only the slots that have to be recomputed start from zero: one index_fill over their indices, then the
                    # table pass
                    self.slots[dev_key].index_fill_(0, slot_idx, 0)
                    check(L.gga_absmax_table(F._p(amax_t), n_amax, n_blocks, F._p(self.slots[dev_key]), 0, F._stream()), 'gga_absmax_table')
                check(L.gga_pack_weights_table(F._p(pack_t), n_pack, total, planes, F._stream()), 'gga_pack_weights_table')
            for op in ops:
                op.versions = [v._version for v, _, _ in op.sources]
                op.epoch = self.epoch
                if VERIFY:
                    op.digest = self._digest(op)
        self.refreshes += 1
        if stale and stale[0].packed.is_cuda:
            dev = stale[0].packed.device
            self.event = torch.cuda.Event()
            cur = torch.cuda.current_stream(dev)
            self.event.record(cur)
            self.stream = cur.cuda_stream

    @staticmethod
    def _digest(op):
        return torch.stack([torch.stack((v.double().sum(), v.double().abs().sum())) for v, _, _ in op.sources]).cpu()

    def _verify(self, op):
        """GGA_BANK_VERIFY=1: the operand was packed from the sources as they are now (sum and L1 of every source)."""
        now = self._digest(op)
        if op.digest is None or not torch.equal(now, op.digest):
            raise RuntimeError('weight bank: a packed operand is stale - its weight was written without bumping the version counter '
                               '(param.data write / raw-pointer kernel): call gga_amd.weight_bank.BANK.invalidate() after such updates')

    def _build_tables(self, ops, dev, planes, need):
        L = _lib.lib()
        entries, first = [], 0
        for op in ops:
            for v, c0, col0 in op.sources:
                k0, k1, n_c, n_col = v.shape
                s = v.stride()
                entries.append(PackEntry(v.data_ptr(), op.packed.data_ptr(), op.slot.data_ptr() if op.slot is not None else None,
                                         s[0], s[1], s[2], s[3], first, k1, k0 * k1, n_c, n_col, c0, col0, op.n_in,
                                         _padded_columns(op.n_out), op.layout, int(op.reverse)))
                first += k0 * k1 * n_c * n_col
        amax, n_blocks, seen = [], 0, set()
        if planes == 2:
            slots = self.slots[str(dev)]
            for op in ops:
                for ptr, n, i in op.blocks:
                    if (ptr, i) in seen or i not in need:
                        continue
                    seen.add((ptr, i))
                    amax.append(AmaxEntry(ptr, slots.data_ptr() + 4 * i, n, n_blocks))
                    n_blocks += int(L.gga_absmax_table_blocks(n))

        def upload(items, ctype):
            if not items:
                return torch.zeros(8, dtype=torch.uint8, device=dev)
            arr = (ctype * len(items))(*items)
            host = torch.from_numpy(np.frombuffer(arr, dtype=np.uint8).copy())
            return host.to(dev)
        slot_idx = torch.tensor(sorted(need) or [0], dtype=torch.long).to(dev)
        return upload(entries, PackEntry), len(entries), first, upload(amax, AmaxEntry), len(amax), n_blocks, slot_idx


BANK = WeightBank()

# every optimizer of the process tells the bank after its step (torch's fused optimizers do not bump the version counters)
from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_STEP_HOOK = _register_step_hook(lambda *_: BANK.invalidate())


def dense_operand(weight, backward, transposed, planes, c0=None):
    """Operand of the 3x3 stride-1 kernels for a Conv2d weight [cout, cin, 3, 3] (any strides): forward = [ky, kx, cin, cout];
    backward-data = the cout -> cin convolution with the taps reversed; ``transposed``: the tile walk with ky and kx exchanged;
    ``c0``: the operand of the output slice [c0, c0 + 128) (scaled like the whole weight)."""
    full = weight.detach()
    w = full if c0 is None else (full[:, c0:c0 + 128] if backward else full[c0:c0 + 128])
    a, b = (3, 2) if transposed else (2, 3)
    view = w.permute(a, b, 0, 1) if backward else w.permute(a, b, 1, 0)
    return BANK.operand([(view, 0, 0)], view.shape[2], view.shape[3], LAYOUT_DENSE, backward, planes, scale_of=[full])


def dense_operand_cat(weights, backward, transposed, planes):
    """The same for the concatenation of ``weights`` along their OUTPUT channels (dim 0) without making it: forward -> the
    columns of one operand side by side (<= 128 in all); backward-data -> its input channels one after the other."""
    a, b = (3, 2) if transposed else (2, 3)
    sources, off = [], 0
    for w in weights:
        w = w.detach()
        if backward:
            sources.append((w.permute(a, b, 0, 1), off, 0))
        else:
            sources.append((w.permute(a, b, 1, 0), 0, off))
        off += w.shape[0]
    cin = weights[0].shape[1]
    n_in, n_out = (off, cin) if backward else (cin, off)
    return BANK.operand(sources, n_in, n_out, LAYOUT_DENSE, backward, planes)


def gather_operand(view, planes, col0=None, col1=None):
    """Operand of the gather-GEMM kernels from a VIEW ``[k0, k1, cin, cout]`` (or ``[kvol, cin, cout]``) of a parameter -
    W[k][c][col] -, or of its columns [col0, col1) (<= 128; scaled like the whole weight). The view must belong to a tensor
    that lives across steps: a temporary would add an operand per step."""
    v = view.detach()
    if v.dim() == 3:
        v = v.unsqueeze(0)
    part = v if col0 is None else v[..., col0:col1]
    return BANK.operand([(part, 0, 0)], part.shape[2], part.shape[3], LAYOUT_GATHER, False, planes, scale_of=[v])
