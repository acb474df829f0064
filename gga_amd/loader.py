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


def collate(batch, samples_per_gpu=1):
    """mmcv.parallel ``collate``: samples (dicts / sequences of ``DataContainer`` or plain values) -> one container per key
    whose payload is a list with one entry per ``samples_per_gpu`` chunk."""
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


def to_step_inputs(collated, device=None, chunk=0, non_blocking=True):
    """A collated batch -> the keyword inputs of the detector's ``forward_train`` (or ``forward_test``) for this rank's
    device: chunk ``chunk`` of every container, ``points`` on ``device`` (uploads from pinned memory are asynchronous:
    announce them with ``Runner.inputs_ready``), everything else as it left the pipeline."""
    out = {}
    for key, value in collated.items():
        value = _take_chunk(value, chunk)
        if device is not None and key in DEVICE_KEYS:
            value = _to_device(value, device, non_blocking)
        out[key] = value
    return out


# ------------------------------------------------------------------------------------------------------------------ loader
def build_dataloader(dataset, samples_per_gpu, workers_per_gpu, num_gpus=1, dist=True, shuffle=True, seed=None,
                     runner_type='EpochBasedRunner', persistent_workers=False, rank=None, world_size=None, worker_seed=None,
                     **kwargs):
    """mmdet 2.x ``build_dataloader`` for the epoch-based runner: per process ``samples_per_gpu`` frames per batch when
    distributed (one process per GPU); group samplers when shuffling. ``seed`` seeds the sampler's order and must be equal on
    all ranks (each takes its block of one shuffled sequence); ``worker_seed`` (default: ``seed``) seeds the loader workers'
    augmentation generators and may differ per rank. One process drives one device: the reference's non-distributed
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
                      collate_fn=partial(collate, samples_per_gpu=samples_per_gpu), pin_memory=kwargs.pop('pin_memory', False),
                      worker_init_fn=init_fn, **kwargs)
