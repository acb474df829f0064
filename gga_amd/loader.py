"""From a dataset to the inputs of ``Runner.step``: the data side of ``train_detector``
(mmdet3d/apis/train.py:205-219 ``build_mmdet_dataloader(ds, samples_per_gpu, workers_per_gpu, num_gpus, dist, seed,
runner_type, persistent_workers)``, :283-285 ``DistSamplerSeedHook``) and of ``build_dataset``
(mmdet3d/datasets/builder.py:29-47).

Everything the reference calls here is third-party and absent from the tree (mmdet 2.x ``build_dataloader`` and its samplers,
mmcv ``collate`` / ``DataContainer`` / ``scatter``): their published behaviour is restated - **parity unpinned** - and the
pieces that decide WHICH frames a rank sees are covered by tests (tests/test_loader.py: every index exactly once per epoch
over the ranks up to the padding, disjoint shards, a new order per epoch, equal order for equal seeds).

* ``RepeatDataset``: the wrapper the shipped config puts around the KITTI dataset (``times`` passes per epoch).
* samplers: with ``shuffle`` the group samplers of mmdet (frames are grouped by the dataset's ``flag`` - all zero for the
  LiDAR datasets - and dealt in whole per-GPU batches, so that every batch holds one group only); ``GroupSampler`` draws from
  numpy's global generator, ``DistributedGroupSampler`` from a generator seeded with ``seed + epoch`` on every rank, each rank
  taking its contiguous block of the shuffled batch sequence; without ``shuffle`` a plain strided ``DistributedSampler``.
* ``collate``: mmcv's rules for ``DataContainer`` payloads (``cpu_only`` and unstacked payloads become per-batch lists,
  stacked ones are padded to a common shape and stacked), applied per ``samples_per_gpu`` chunk.
* ``to_step_inputs``: what ``MMDistributedDataParallel.scatter`` does for the one device of a rank - the chunk of this rank
  is unwrapped - with one deliberate difference: only ``points`` go to the device. The label-side tensors are consumed on the
  host by ``CenterHead_GGA.pack_targets`` (one packed upload per step); moving them to the device first would cost an upload
  each and a synchronising read back."""
import math
from collections.abc import Mapping, Sequence
from functools import partial

import numpy as np
import collections

import torch
from torch.utils.data import DataLoader, Sampler
from torch.utils.data.dataloader import default_collate

from .pipelines import DataContainer
from .registry import DATASETS, build_from_cfg


# ---------------------------------------------------------------------------------------------------------------- datasets
@DATASETS.register_module()
class RepeatDataset:
    """``times`` passes over ``dataset`` per epoch (mmdet ``RepeatDataset``): item i is item ``i % len(dataset)``."""

    def __init__(self, dataset, times):
        self.dataset, self.times = dataset, int(times)
        self.CLASSES = getattr(dataset, 'CLASSES', None)
        if hasattr(dataset, 'flag'):
            self.flag = np.tile(dataset.flag, self.times)
        self._ori_len = len(dataset)

    def __getitem__(self, idx):
        return self.dataset[idx % self._ori_len]

    def __len__(self):
        return self.times * self._ori_len


def build_dataset(cfg, default_args=None):
    """Config dict -> dataset: a list of configs is concatenated, ``RepeatDataset`` wraps the dataset it names, anything else
    is looked up in ``DATASETS``."""
    if isinstance(cfg, (list, tuple)):
        from torch.utils.data import ConcatDataset
        return ConcatDataset([build_dataset(c, default_args) for c in cfg])
    if cfg['type'] == 'RepeatDataset':
        return RepeatDataset(build_dataset(cfg['dataset'], default_args), cfg['times'])
    if cfg['type'] in ('ConcatDataset', 'ClassBalancedDataset', 'CBGSDataset'):
        raise NotImplementedError(f'{cfg["type"]} is not used by configs/gga')
    return build_from_cfg(cfg, DATASETS, default_args)


def group_flags(dataset):
    flag = getattr(dataset, 'flag', None)
    return np.zeros(len(dataset), dtype=np.uint8) if flag is None else np.asarray(flag)


# ---------------------------------------------------------------------------------------------------------------- samplers
def _padded_to(indices, length, rng_choice=None):
    """``indices`` repeated cyclically (or topped up by ``rng_choice``) up to ``length`` entries."""
    if len(indices) >= length:
        return indices[:length]
    if rng_choice is not None:
        return np.concatenate([indices, rng_choice(indices, length - len(indices))])
    reps = -(-length // len(indices))
    return np.tile(indices, reps)[:length]


class GroupSampler(Sampler):
    """Single-process shuffling (mmdet ``GroupSampler``): every group is shuffled with numpy's global generator, topped up by
    random repeats to a multiple of ``samples_per_gpu``, cut into batches, and the batches are shuffled."""

    def __init__(self, dataset, samples_per_gpu=1):
        self.flag = group_flags(dataset).astype(np.int64)
        self.samples_per_gpu = int(samples_per_gpu)
        sizes = np.bincount(self.flag)
        self.num_samples = int(sum(-(-int(s) // self.samples_per_gpu) * self.samples_per_gpu for s in sizes))

    def __iter__(self):
        spg = self.samples_per_gpu
        batches = []
        for g, size in enumerate(np.bincount(self.flag)):
            if size == 0:
                continue
            members = np.where(self.flag == g)[0]
            np.random.shuffle(members)
            members = _padded_to(members, -(-len(members) // spg) * spg, np.random.choice)
            batches.append(members.reshape(-1, spg))
        batches = np.concatenate(batches)
        order = np.random.permutation(len(batches))
        return iter(batches[order].reshape(-1).astype(np.int64).tolist())

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler(Sampler):
    """mmdet ``DistributedGroupSampler``: the same on every rank from ``torch.Generator().manual_seed(seed + epoch)`` - groups
    shuffled, repeated cyclically up to a multiple of ``samples_per_gpu * num_replicas``, batches shuffled - and rank r takes
    the r-th contiguous block of ``num_samples`` entries. ``set_epoch`` before every epoch (``DistSamplerSeedHook``)."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None, seed=0):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.flag = group_flags(dataset).astype(np.int64)
        self.samples_per_gpu, self.num_replicas, self.rank = int(samples_per_gpu), int(num_replicas), int(rank)
        self.seed = 0 if seed is None else int(seed)
        self.epoch = 0
        per_round = self.samples_per_gpu * self.num_replicas
        self.group_total = [-(-int(s) // per_round) * per_round for s in np.bincount(self.flag)]
        self.total_size = int(sum(self.group_total))
        self.num_samples = self.total_size // self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __iter__(self):
        gen = torch.Generator()
        gen.manual_seed(self.epoch + self.seed)
        parts = []
        for g, total in enumerate(self.group_total):
            if total == 0:
                continue
            members = np.where(self.flag == g)[0]
            members = members[torch.randperm(len(members), generator=gen).numpy()]
            parts.append(_padded_to(members, total))
        batches = np.concatenate(parts).reshape(-1, self.samples_per_gpu)
        batches = batches[torch.randperm(len(batches), generator=gen).numpy()]
        mine = batches.reshape(-1)[self.num_samples * self.rank:self.num_samples * (self.rank + 1)]
        return iter(mine.astype(np.int64).tolist())

    def __len__(self):
        return self.num_samples


class DistributedSampler(Sampler):
    """mmdet ``DistributedSampler``: (optionally shuffled, ``seed + epoch``) index list repeated up to a multiple of the
    number of ranks, rank r taking every ``num_replicas``-th entry from r."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True, seed=0):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.n, self.num_replicas, self.rank, self.shuffle = len(dataset), int(num_replicas), int(rank), shuffle
        self.seed = 0 if seed is None else int(seed)
        self.epoch = 0
        self.num_samples = -(-self.n // self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __iter__(self):
        if self.shuffle:
            gen = torch.Generator()
            gen.manual_seed(self.epoch + self.seed)
            order = torch.randperm(self.n, generator=gen).numpy()
        else:
            order = np.arange(self.n)
        order = _padded_to(order, self.total_size)
        return iter(order[self.rank:self.total_size:self.num_replicas].astype(np.int64).tolist())

    def __len__(self):
        return self.num_samples


def worker_init_fn(worker_id, num_workers, rank, seed):
    """Every loader worker of every rank gets its own seed for numpy / random / torch (the augmentations draw from them)."""
    import random
    worker_seed = num_workers * rank + worker_id + seed
    np.random.seed(worker_seed)
    random.seed(worker_seed)
    torch.manual_seed(worker_seed)


# ----------------------------------------------------------------------------------------------------------------- collate
def _stack_padded(chunk):
    """Stacked ``DataContainer`` payloads of one per-GPU chunk: the last ``pad_dims`` dimensions are padded (at their end,
    with the container's ``padding_value``) to the largest size in the chunk, then stacked."""
    first = chunk[0]
    if first.pad_dims is None:
        return default_collate([s.data for s in chunk])
    nd, pd = first.data.dim(), first.pad_dims
    assert nd > pd, 'pad_dims must leave a leading dimension'
    for s in chunk:
        assert tuple(s.data.shape[:nd - pd]) == tuple(first.data.shape[:nd - pd]), 'leading dimensions must agree'
    target = [max(s.data.size(-d) for s in chunk) for d in range(1, pd + 1)]          # last dimension first
    padded = []
    for s in chunk:
        pad = []
        for d in range(1, pd + 1):
            pad += [0, target[d - 1] - s.data.size(-d)]
        padded.append(torch.nn.functional.pad(s.data, pad, value=s.padding_value))
    return default_collate(padded)


def collate(batch, samples_per_gpu=1, packed=False):
    """mmcv.parallel ``collate``: samples (dicts / sequences of ``DataContainer`` or plain values) -> one container per key
    whose payload is a list with one entry per ``samples_per_gpu`` chunk. ``packed``: every chunk's per-frame list as one
    ``PackedFrames`` (``pack_collated``; ``to_step_inputs`` undoes it)."""
    if packed:
        out = collate(batch, samples_per_gpu)
        return pack_collated(out) if isinstance(out, dict) else out
    if not isinstance(batch, Sequence):
        raise TypeError(f'{type(batch)} is not a sequence of samples')
    head = batch[0]
    if isinstance(head, DataContainer):
        chunks = [batch[i:i + samples_per_gpu] for i in range(0, len(batch), samples_per_gpu)]
        if head.stack and not head.cpu_only:
            return DataContainer([_stack_padded(c) for c in chunks], True, head.padding_value)
        return DataContainer([[s.data for s in c] for c in chunks], head.stack, head.padding_value, cpu_only=head.cpu_only)
    if isinstance(head, Mapping):
        return {key: collate([sample[key] for sample in batch], samples_per_gpu) for key in head}
    if isinstance(head, Sequence) and not isinstance(head, (str, bytes)):
        return [collate(list(column), samples_per_gpu) for column in zip(*batch)]
    return default_collate(batch)


DEVICE_KEYS = ('points',)


# ------------------------------------------------------------------------------------------------------- packed hand-over
# A collated batch of 16 GGA frames holds ~450 tensors (per frame: points, labels, four GGA arrays, the boxes, and one in-box
# point set per object). Sent from a loader worker as they are, each travels as its own shared-memory segment and the train
# process spends ~20 ms per batch reopening them - on the thread that launches the step. ``collate(..., packed=True)`` (what
# ``build_dataloader`` installs) concatenates every per-frame list of a chunk into ONE tensor plus its split sizes inside the
# worker; ``to_step_inputs`` hands the detector the per-frame views again - the same lists ``forward_train`` takes from the
# reference's collate, as ``FrameList`` objects that also remember the flat tensor, which ``CenterHead_GGA.pack_targets``
# and the voxelizer use instead of concatenating the views once more.
class PackedFrames:
    """One chunk's per-frame list of tensors (``sizes`` rows each) - or list of lists of tensors (``inner``: the number of
    tensors of each frame, ``sizes`` then counts the rows of every inner tensor) - as one tensor. Picklable payload."""

    __slots__ = ('_flat', 'sizes', 'inner', 'boxes', 'arena', 'where')

    def __init__(self, flat, sizes, inner=None, boxes=None):
        self._flat, self.sizes, self.inner, self.boxes = flat, sizes, inner, boxes
        self.arena = self.where = None          # `into_arena`: the rows live in a byte buffer shared by the whole batch

    @property
    def flat(self):
        if self._flat is None:
            offset, nbytes, dtype, shape = self.where
            self._flat = self.arena[offset:offset + nbytes].view(dtype).view(shape)
        return self._flat

    def __getstate__(self):
        if self.arena is not None:              # (the arena is one object for all payloads of a batch: pickled once)
            return (None, self.sizes, self.inner, self.boxes, self.arena, self.where)
        return (self._flat, self.sizes, self.inner, self.boxes, None, None)

    def __setstate__(self, state):
        self._flat, self.sizes, self.inner, self.boxes, self.arena, self.where = state

    def __len__(self):
        return len(self.sizes if self.inner is None else self.inner)

    # The collate contract: mmcv's collate hands every non-stacked DataContainer payload over as a list with one entry per
    # frame. Code that walks a batch without ``to_step_inputs`` (user hooks, ``batch['points'].data[0][i]``) finds that list
    # here: the per-frame views, made on demand.
    def frames(self):
        return _unpack_frames(self)

    def __iter__(self):
        return iter(self.frames())

    def __getitem__(self, i):
        return self.frames()[i]


class FrameList(list):
    """The per-frame views of a ``PackedFrames`` payload; ``flat`` / ``sizes`` (/ ``inner``) describe the memory behind them."""
    flat = sizes = inner = None


def _pack_frames(frames):
    """list of tensors / list of lists of tensors / list of boxes -> PackedFrames, or None when the list is something else."""
    from .box3d import LiDARInstance3DBoxes
    if not frames:
        return None
    if all(isinstance(f, LiDARInstance3DBoxes) for f in frames):
        if len({(f.box_dim, f.with_yaw) for f in frames}) != 1:
            return None
        return PackedFrames(torch.cat([f.tensor for f in frames], 0), [len(f.tensor) for f in frames], boxes=(frames[0].box_dim, frames[0].with_yaw))
    if all(torch.is_tensor(f) for f in frames):
        if len({(f.dtype, tuple(f.shape[1:])) for f in frames}) != 1 or frames[0].dim() == 0:
            return None
        return PackedFrames(torch.cat(frames, 0), [int(f.shape[0]) for f in frames])
    if all(isinstance(f, (list, tuple)) and all(torch.is_tensor(t) for t in f) for f in frames):
        inner = [t for f in frames for t in f]
        if not inner or len({(t.dtype, tuple(t.shape[1:])) for t in inner}) != 1 or inner[0].dim() == 0:
            return None
        return PackedFrames(torch.cat(inner, 0), [int(t.shape[0]) for t in inner], inner=[len(f) for f in frames])
    return None


def into_arena(payloads, align=64):
    """Moves the rows of all ``payloads`` (``PackedFrames``) into ONE byte tensor: a batch then crosses the process boundary as a
    single shared-memory segment - every segment costs the receiving (train) thread a descriptor hand-over and an mmap, ~0.5 ms
    each - and the payloads keep (offset, bytes, dtype, shape) views into it."""
    if not payloads:
        return None
    total, where = 0, []
    for p in payloads:
        flat = p._flat.contiguous()
        nbytes = flat.numel() * flat.element_size()
        where.append((total, nbytes, flat.dtype, tuple(flat.shape), flat))
        total += -(-max(nbytes, 1) // align) * align
    arena = torch.empty(total, dtype=torch.uint8)
    for p, (offset, nbytes, dtype, shape, flat) in zip(payloads, where):
        if nbytes:
            arena[offset:offset + nbytes].copy_(flat.reshape(-1).view(torch.uint8))
        p.arena, p.where, p._flat = arena, (offset, nbytes, dtype, shape), None
    return arena


def pack_collated(collated, arena=True):
    """Replaces the per-frame lists inside a collated batch's containers by ``PackedFrames`` (in place; what cannot be packed
    - stacked payloads, meta dicts - stays); ``arena``: all of them in one byte buffer (``into_arena``)."""
    made = []
    for key, value in collated.items():
        if isinstance(value, DataContainer) and not value.stack and isinstance(value.data, list):
            for c, chunk in enumerate(value.data):
                packed = _pack_frames(chunk) if isinstance(chunk, list) else None
                if packed is not None:
                    value.data[c] = packed
                    made.append(packed)
    if arena:
        into_arena(made)
    return collated


class PointUploader:
    """Host -> device hand-over of a batch's packed points for the train loop: a ring of ``DEPTH`` pinned host buffers and
    device buffers and an upload stream of its own.

    A loader batch arrives in pageable (shared) memory. ``tensor.to(device, non_blocking=True)`` from pageable memory is a
    staged copy that the runtime orders on the CURRENT stream and waits for on the host: the train thread - which runs a step or
    two ahead of the device - would stop until the device has caught up, every step. Here the rows are copied into a pinned
    buffer (a host memcpy, ~1 ms for 16 x 20 k points) and uploaded from there on the upload stream, which does not wait for
    the step that is running: slot k % DEPTH was last read by the step DEPTH batches ago, and the upload only waits for the
    work that was queued before the PREVIOUS call (which includes that step). The caller's stream then waits for the upload's
    event, so everything queued afterwards (the step itself, the prefetch stream that forks from it) sees the points."""

    DEPTH = 3

    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.pinned, self.dev, self.copied = [None] * self.DEPTH, [None] * self.DEPTH, [None] * self.DEPTH
        self.queued = []            # events on the caller's stream, one per call: the work queued before that call
        self.calls = 0
        # Loops that report their steps (``step_done`` after queueing the step that consumed the OLDEST uploaded batch;
        # ``Runner.train_epochs`` does) give every slot the event of its last reader, and the next upload into the slot waits
        # for exactly that - whatever the loop's look-ahead. Loops that never report keep the call-order rule below, which is
        # only safe when at most one batch is uploaded ahead of the step being queued.
        self.readers = [None] * self.DEPTH
        self.unread = collections.deque()

    def step_done(self):
        """The step that reads the oldest not-yet-consumed upload has just been queued on the current stream."""
        if self.unread:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.readers[self.unread.popleft()] = ev

    def _buffers(self, slot, like):
        rows = like.shape[0]
        cur = self.pinned[slot]
        if cur is None or cur.shape[0] < rows or cur.shape[1:] != like.shape[1:] or cur.dtype != like.dtype:
            cap = max(int(rows * 1.25) + 1024, 1)
            self.pinned[slot] = torch.empty((cap,) + tuple(like.shape[1:]), dtype=like.dtype).pin_memory()
            self.dev[slot] = torch.empty((cap,) + tuple(like.shape[1:]), dtype=like.dtype, device=self.device)
            self.dev[slot].record_stream(self.stream)
        return self.pinned[slot], self.dev[slot]

    def __call__(self, flat):
        slot = self.calls % self.DEPTH
        self.calls += 1
        main = torch.cuda.current_stream(self.device)
        now = torch.cuda.Event()
        now.record(main)
        self.queued.append(now)
        if self.copied[slot] is not None:
            self.copied[slot].synchronize()          # the pinned buffer's previous upload (DEPTH calls ago: long done)
        pinned, dev = self._buffers(slot, flat)
        n = flat.shape[0]
        pinned[:n].copy_(flat)
        if slot in self.unread:                      # the slot's previous content was never reported as consumed: everything queued so far may read it
            self.stream.wait_event(now)
            self.unread.remove(slot)
        elif self.readers[slot] is not None:
            self.stream.wait_event(self.readers[slot])
        elif len(self.queued) >= 2:
            self.stream.wait_event(self.queued[-2])  # the slot's previous reader was queued before the previous call
        self.readers[slot] = None
        self.unread.append(slot)
        del self.queued[:-2]
        with torch.cuda.stream(self.stream):
            dev[:n].copy_(pinned[:n], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.copied[slot] = done
        main.wait_event(done)
        return dev[:n]


def _unpack_frames(p, device=None, non_blocking=True, uploader=None):
    if device is None:
        flat = p.flat
    elif uploader is not None and not p.flat.is_cuda:
        flat = uploader(p.flat)
    else:
        flat = p.flat.to(device, non_blocking=non_blocking)
    views = list(torch.split(flat, p.sizes, 0)) if len(p.sizes) else []
    out = FrameList()
    if p.boxes is not None:
        from .box3d import LiDARInstance3DBoxes
        box_dim, with_yaw = p.boxes
        for v in views:
            b = LiDARInstance3DBoxes.__new__(LiDARInstance3DBoxes)        # (the constructor would clone the view)
            b.tensor, b.box_dim, b.with_yaw = v, box_dim, with_yaw
            out.append(b)
    elif p.inner is not None:
        i = 0
        for n in p.inner:
            out.append(views[i:i + n])
            i += n
    else:
        out.extend(views)
    out.flat, out.sizes, out.inner = flat, p.sizes, p.inner
    return out


def chunks_in(collated):
    """Number of per-GPU chunks in a collated batch."""
    for value in collated.values():
        while isinstance(value, (list, tuple)) and value:
            value = value[0]
        if isinstance(value, DataContainer):
            return len(value.data)
    return 1


def _take_chunk(value, chunk):
    if isinstance(value, DataContainer):
        return value.data[chunk]
    if isinstance(value, (list, tuple)):          # test-time samples: a list over augmentations of containers
        return [_take_chunk(v, chunk) for v in value]
    return value


def _to_device(value, device, non_blocking):
    if isinstance(value, (list, tuple)):
        return [_to_device(v, device, non_blocking) for v in value]
    return value.to(device, non_blocking=non_blocking) if torch.is_tensor(value) else value


def to_step_inputs(collated, device=None, chunk=0, non_blocking=True, uploader=None):
    """A collated batch -> the keyword inputs of the detector's ``forward_train`` (or ``forward_test``) for this rank's
    device: chunk ``chunk`` of every container, ``points`` on ``device`` (uploads from pinned memory are asynchronous:
    announce them with ``Runner.inputs_ready``), everything else as it left the pipeline. ``uploader``: a ``PointUploader``
    that takes packed points to the device without stalling the calling thread (``Runner.train_epochs`` passes one)."""
    out = {}
    for key, value in collated.items():
        value = _take_chunk(value, chunk)
        on_device = device is not None and key in DEVICE_KEYS
        if isinstance(value, PackedFrames):                 # one upload for the whole chunk, then per-frame views
            value = _unpack_frames(value, device if on_device else None, non_blocking, uploader)
        elif on_device:
            value = _to_device(value, device, non_blocking)
        out[key] = value
    return out


# ------------------------------------------------------------------------------------------------------------------ loader
def build_dataloader(dataset, samples_per_gpu, workers_per_gpu, num_gpus=1, dist=True, shuffle=True, seed=None,
                     runner_type='EpochBasedRunner', persistent_workers=False, rank=None, world_size=None, worker_seed=None,
                     packed=True, **kwargs):
    """mmdet 2.x ``build_dataloader`` for the epoch-based runner: per process ``samples_per_gpu`` frames per batch when
    distributed (one process per GPU); group samplers when shuffling. ``seed`` seeds the sampler's order and must be equal on
    all ranks (each takes its block of one shuffled sequence); ``worker_seed`` (default: ``seed``) seeds the loader workers'
    augmentation generators and may differ per rank. ``packed`` (default): batches cross the process boundary as one tensor per
    key (``PackedFrames``), unpacked by ``to_step_inputs``. One process drives one device: the reference's non-distributed
    ``num_gpus > 1`` mode (MMDataParallel scattering ``num_gpus`` chunks) does not exist here and is refused."""
    if runner_type != 'EpochBasedRunner':
        raise NotImplementedError('configs/gga train with the EpochBasedRunner')
    if not dist and num_gpus != 1:
        raise NotImplementedError(f'{num_gpus} GPUs without a launcher: one process drives one GPU - start one rank per GPU '
                                  f'(tools/train.py --launcher pytorch)')
    if dist:
        import torch.distributed as td
        rank = td.get_rank() if rank is None else rank
        world_size = td.get_world_size() if world_size is None else world_size
        sampler = (DistributedGroupSampler(dataset, samples_per_gpu, world_size, rank, seed=seed) if shuffle else
                   DistributedSampler(dataset, world_size, rank, shuffle=False, seed=seed))
        batch_size, num_workers = samples_per_gpu, workers_per_gpu
    else:
        rank = 0
        sampler = GroupSampler(dataset, samples_per_gpu) if shuffle else None
        batch_size, num_workers = num_gpus * samples_per_gpu, num_gpus * workers_per_gpu
    worker_seed = seed if worker_seed is None else worker_seed
    init_fn = partial(worker_init_fn, num_workers=num_workers, rank=rank, seed=worker_seed) if worker_seed is not None else None
    if num_workers > 0:
        kwargs['persistent_workers'] = persistent_workers
    return DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                      collate_fn=partial(collate, samples_per_gpu=samples_per_gpu, packed=packed), pin_memory=kwargs.pop('pin_memory', False),
                      worker_init_fn=init_fn, **kwargs)
