"""The test / pseudo-label entry of the GGA recipe (README.md:187-192 of the reference: ``tools/generate_pseudo_labels_gga.py
<matching config> <checkpoint> --eval mAP``): run the trained detector over the dataset in test mode and hand the detections
to the dataset's ``evaluate`` - for ``KittiDataset_GGA_match`` that is the 2D-box matching which writes the pseudo-label
file (tools/generate_pseudo_labels_gga.py:133-262, mmdet3d/apis/test.py:15-90 ``single_gpu_test``).

Third-party pieces restated (parity unpinned): mmcv's ``load_checkpoint`` (``train.Runner.load_checkpoint``'s rules), mmdet's
loader with ``shuffle=False`` (``loader.build_dataloader``), ``MMDataParallel.scatter`` for one device
(``loader.to_step_inputs``)."""
import pickle

import torch

from .loader import build_dataloader, build_dataset, chunks_in, to_step_inputs
from .registry import build_detector


def single_gpu_test(model, data_loader, device=None, progress=None, planes=None):
    """``model(return_loss=False, rescale=True, **batch)`` over the loader -> list with one result dict per frame.
    ``planes``: arithmetic of the matrix kernels for the run (``dense_conv.PLANES``: 2 = the two-fp16-plane form a guarded
    training run stayed on - ``generate_pseudo_labels`` passes what the checkpoint records -, 3 / None = the library default)."""
    from . import dense_conv
    model.eval()
    device = device or next(model.parameters()).device
    results = []
    outer = dense_conv.PLANES
    if planes is not None and not dense_conv.PLANES_PINNED:
        dense_conv.PLANES = int(planes)
    try:
        for batch in data_loader:
            for chunk in range(chunks_in(batch)):      # the loader may deliver several per-GPU chunks: one after the other here
                # test-time layout: every key is a list over augmentations of a list over frames
                data = to_step_inputs(batch, device, chunk)
                with torch.no_grad():
                    out = model(return_loss=False, rescale=True, **data)
                results.extend(out)
                if progress:
                    progress(len(results))
    finally:
        dense_conv.PLANES = outer
    return results


def collect_results(part, size, tmpdir=None, gpu_collect=False):
    """mmdet ``collect_results_cpu`` / ``collect_results_gpu`` (mmdet/apis/test.py; third-party, restated): every rank hands in
    the results of its shard - the strided shard of ``loader.DistributedSampler(shuffle=False)``: rank r holds frames r,
    r + world, ... - and rank 0 gets the ``size`` results of the whole dataset in dataset order (the sampler's padding repeats
    cut off); the other ranks get None. ``gpu_collect``: through ``all_gather_object`` (the process group's transport);
    otherwise through pickle files in ``tmpdir`` (a shared directory; created by rank 0 when not given), as the reference's
    default."""
    import os
    import shutil
    import tempfile
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    own_tmpdir = tmpdir is None
    if gpu_collect:
        parts = [None] * world
        dist.all_gather_object(parts, part)
    else:
        if tmpdir is None:
            name = [tempfile.mkdtemp(prefix='gga_collect_') if rank == 0 else None]
            dist.broadcast_object_list(name, src=0)
            tmpdir = name[0]
        else:
            # a directory the caller named: mmdet removes it whole afterwards - fine for one this run creates, not for one that was
            # there before with other contents: rank 0 looks first, then everybody may write
            if rank == 0:
                own_tmpdir = not os.path.isdir(tmpdir)
                os.makedirs(tmpdir, exist_ok=True)
            dist.barrier()
        with open(os.path.join(tmpdir, f'part_{rank}.pkl'), 'wb') as f:
            pickle.dump(part, f)
        dist.barrier()
        parts = None
        if rank == 0:
            parts = [pickle.load(open(os.path.join(tmpdir, f'part_{r}.pkl'), 'rb')) for r in range(world)]
            if own_tmpdir:
                shutil.rmtree(tmpdir)
            else:                              # a directory that existed before this run: only this run's files leave it
                for r in range(world):
                    os.remove(os.path.join(tmpdir, f'part_{r}.pkl'))
    if rank != 0:
        return None
    # every rank holds the same number of results (the sampler pads the shards to equal length): a short part - a rank that
    # lost frames on an error path - must not silently truncate the interleave below
    per_rank = -(-size // world)
    assert all(len(p) == per_rank for p in parts), f'results per rank {[len(p) for p in parts]}, expected {per_rank} each'
    ordered = []
    for group in zip(*parts):               # frame i of the dataset is entry i // world of rank i % world
        ordered.extend(group)
    return ordered[:size]


def multi_gpu_test(model, data_loader, tmpdir=None, gpu_collect=False, device=None, progress=None, planes=None):
    """mmdet ``multi_gpu_test`` (tools/generate_pseudo_labels_gga.py:242 of the reference, started per GPU by
    tools/dist_pseudo.sh:11-22): every rank tests its shard of the loader, rank 0 returns the results of the whole dataset in
    dataset order."""
    part = single_gpu_test(model, data_loader, device, progress, planes)
    return collect_results(part, len(data_loader.dataset), tmpdir, gpu_collect)


def load_weights(model, filename, map_location='cpu', strict=False):
    """Checkpoint file -> model weights (``state_dict`` entry or a bare state dict; a leading ``module.`` dropped). -> the
    file's ``meta`` dict."""
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    state = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    state = {(k[7:] if k.startswith('module.') else k): v for k, v in state.items()}
    model.load_state_dict(state, strict=strict)
    return ckpt.get('meta', {}) if isinstance(ckpt, dict) else {}


def generate_pseudo_labels(cfg, checkpoint, out=None, eval_metrics=('mAP',), eval_options=None, device='cuda:0', model=None,
                           channels_last=True, progress=None, distributed=False, tmpdir=None, gpu_collect=False):
    """The flow of the reference's tool for one process: dataset ``cfg.data.test`` in test mode, loader
    (``cfg.data.test_dataloader`` over the defaults samples_per_gpu=1, workers_per_gpu=2, no shuffling), detector from
    ``cfg.model`` with the checkpoint's weights (``model``: an already built detector instead), ``single_gpu_test``, the raw
    outputs pickled to ``out`` when given, then ``dataset.evaluate(outputs, metric=..., **eval_options)`` - the matching
    dataset writes the pseudo-label file there (``pseudo_label_file`` in ``eval_options`` names it). -> (outputs, what
    ``evaluate`` returned or None). ``distributed`` (one process per GPU, process group initialised - ``train.init_dist``):
    the frames are sharded over the ranks (``multi_gpu_test``), rank 0 collects, writes and evaluates; the other ranks
    return (None, None)."""
    device = torch.device(device)
    test_cfg = cfg.data['test']
    test_cfg['test_mode'] = True
    dataset = build_dataset(test_cfg)
    loader_cfg = dict(samples_per_gpu=1, workers_per_gpu=2, dist=bool(distributed), shuffle=False)
    loader_cfg.update(cfg.data.get('test_dataloader', {}))
    loader = build_dataloader(dataset, **loader_cfg)
    planes = None
    if model is None:
        cfg.model['train_cfg'] = None
        mcfg = cfg.model
        if channels_last and mcfg.get('pts_middle_encoder', {}).get('type') in ('PointPillarsScatter', 'SparseEncoder'):
            mcfg['pts_middle_encoder']['channels_last'] = True
        model = build_detector(mcfg, test_cfg=cfg.get('test_cfg'))
        meta = load_weights(model, checkpoint)
        model.CLASSES = meta.get('CLASSES', dataset.CLASSES)
        # the arithmetic the run trained on (train.Runner.save_checkpoint records it): a model whose guarded training stayed on
        # two fp16 planes is evaluated on them; anything else on the library default
        if meta.get('gga_amd_planes') == 2 and not meta.get('gga_amd_fell_back'):
            planes = 2
        model = model.to(device)
        if channels_last:
            from .cnn import to_channels_last
            model = to_channels_last(model)
    if distributed:
        outputs = multi_gpu_test(model, loader, tmpdir, gpu_collect, device, progress, planes)
        if outputs is None:                    # not rank 0
            return None, None
    else:
        outputs = single_gpu_test(model, loader, device, progress, planes)
    if out:
        if not out.endswith(('.pkl', '.pickle')):
            raise ValueError('The output file must be a pkl file.')
        with open(out, 'wb') as f:
            pickle.dump(outputs, f)
    result = None
    if eval_metrics:
        kwargs = dict(cfg.get('evaluation', {}))
        for key in ('interval', 'tmpdir', 'start', 'gpu_collect', 'save_best', 'rule'):          # hook arguments, not evaluate()'s
            kwargs.pop(key, None)
        kwargs.update(dict(metric=list(eval_metrics), **(eval_options or {})))
        kwargs.setdefault('device', str(device))
        result = dataset.evaluate(outputs, **kwargs)
    return outputs, result
