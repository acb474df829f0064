#!/usr/bin/env python
"""Headline benchmark: GGA train step on synthetic KITTI-shaped frames (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          # any N: starts its own ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full train step of BASELINE config #2 on one batch: batched hard voxelize
(PointPillars grid) -> PillarFeatureNet -> pillar scatter -> SECOND -> SECONDFPN ->
CenterHead_GGA -> GGA losses -> backward -> (DDP all-reduce) -> grad clip -> AdamW, bs=16
frames per GPU, fp32. Point clouds are resident in HBM before the timed region; weak scaling
(every rank steps its own 16 frames, gradients all-reduced over RCCL).

Launch (reference: tools/dist_train.sh:10-20 wraps tools/train.py in torch.distributed.launch,
one process per GPU; mmdet3d/apis/train.py:222-231 wraps the model in DDP): with ``--gpus N > 1``
and no RANK in the environment this process starts the N ranks itself as child processes of
``torch.distributed.run`` BEFORE touching the GPU, relays rank 0's JSON line and exits with the
children's status. With fewer visible devices than ranks the ranks share devices over gloo (a
functional check, reported as such in ``config.backend``).

Rank 0 prints ONE JSON line; ``roofline`` is the pillar scatter (BASELINE's HBM metric),
``mfma_roofline`` the 128 -> 128 dense 3x3 convolutions of SECOND's second stage (the dominant instantiation of the dense
kernel; its share of the step is in the object), both timed with
HIP events inside the timed steps; ``second_trunk`` is the reference's shipped model
(configs/gga/gga_kitti_config.py: sparse-conv trunk, BASELINE config #3's per-GPU workload, bs 8)
and ``pgd_trunk`` the camera-only retraining model (configs/gga/gga_pdg.py, BASELINE config #5, bs 12)
stepped in the same run; ``cpu_baseline`` is the oracle's CPU restatement of the same step on a
bounded sample plus its pieces (N=1 only).
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 achievable
BF16_PEAK_TFLOPS = 2500.0
ARITH = ('fp32 in / fp32 out, fp32 accumulate. Every convolution (dense 3x3, strided, transposed, sparse 3D; forward, '
         'backward-data, weight gradient): operands scaled by a power of two to their largest finite magnitude and split into '
         'two round-to-nearest fp16 planes (22-bit significands), three f16 MFMA partial products, exact rescale; the sparse '
         'kernels sum every offset\'s products as an accumulator chain of its own and add the 27 partial results once (the '
         'reference\'s gather -> GEMM -> scatter-add order; round 6). RMS error vs float64 on the operands of real steps, every '
         'convolution shape and direction of both LiDAR configs (tests/test_precision_gpu.py, profiles/r06_precision_shapes.json): '
         'dense shapes 0.72 x torch-CPU\'s and 0.73 x MIOpen\'s fp32 in the median, never above 1.34 x the less accurate of the two; '
         'sparse shapes 0.7-1.2 x a per-offset fp32 sgemm + scatter-add (the 4 / 16-channel levels, where the sgemm\'s chain is 4-16 terms: 1.1-1.8 x; '
         'asserted: <= 2 x; round 5: 1.5-5.1 x); per element an absolute accuracy of 2^-40 of its tensor\'s largest magnitude. '
         'Selected by the train Runner under its range guard (every operand of iteration 0 and of every 500th iteration is '
         'measured; an operand with > 0.1% of its non-zero elements below 2^-30 of its maximum sends the run to the library '
         'default: three bf16 planes / six products, fp32\'s full exponent range - timed in `planes3`). Head output convs: fp32 MFMA. '
         'Everything else plain fp32')
# `dtype`: what the matrix kernels compute in. fp32 tensors in and out and fp32 accumulation, but the multiplications run on
# 16-bit operand planes - not plain fp32, so the field does not say "f32":
DTYPE_PLANES2 = ('f32 tensors; products on 2 x f16 operand planes (block-scaled), f32 accumulate - measured no less accurate than fp32 kernels: '
                 'per convolution 0.7 x (dense) / 0.7-1.2 x (sparse) the RMS error of an fp32 FMA chain vs float64, whole-step losses over 8 seeds '
                 '<= 2.5e-5 from float64 (fp32 CPU step: <= 1.6e-4), see `arith` / `parity` / `parity_at_bench_size`')
DTYPE_PLANES3 = 'f32 tensors, products on 3 x bf16 operand planes (24-bit significands, fp32 range), f32 accumulate'
PARITY = ('tests/test_model_gpu.py: all 18 losses of a whole step within 1e-4 of the float64 step AND of the fp32 CPU step (plus that '
          'step\'s own distance from float64) for this arithmetic, on both LiDAR configs - PointPillars on seeds 1, 2, 3; gga_kitti_config.py '
          '= BASELINE config 1 (4 frames of 20 k points) on EIGHT seeds (weights and frames differ per seed; over them the GPU step is '
          '7e-6..2.5e-5 from float64, the fp32 CPU step 7e-6..1.6e-4) - and at the bench size (16 / 8 frames x 20 k points) against the fp32 '
          'CPU step; tests/test_trained_regime_gpu.py: train_detector for 320 optimizer steps on an on-disk tree, every 50th step re-synced '
          'against the fp32 CPU step from the same weights (worst of 18 losses <= 2.2e-6), the final step against float64 and fp32 on both '
          'forms; `parity_at_bench_size` in this line: the same comparison for the batch and weights timed here; tests/test_precision_gpu.py '
          'compares every convolution shape of both configs with an fp32 FMA chain on real step operands')
PP_CONFIG = os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py')
SECOND_CONFIG = os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py')
PGD_CONFIG = os.path.join(REPO, 'configs', 'gga', 'gga_pdg.py')
FCAF3D_CONFIG = os.path.join(REPO, 'configs', 'fcaf3d', 'fcaf3d_8x2_sunrgbd-3d-10class.py')


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='frames per GPU')
    ap.add_argument('--config', default=PP_CONFIG)
    ap.add_argument('--nchw', action='store_true',
                    help='keep the reference NCHW memory layout for the BEV trunk (default: channels-last memory, '
                         'same logical tensors and values)')
    ap.add_argument('--head-init-scale', type=float, default=0.05,
                    help='damp the random init of the regression heads\' output convs (see damp_head_init)')
    ap.add_argument('--miopen-find', action='store_true', help='torch.backends.cudnn.benchmark=True (MIOpen find mode)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--device-warmup-seconds', type=float, default=2.0,
                    help='keep the device busy with generic matrix products for this long before anything is built or timed '
                         '(a fresh box starts at idle clocks: the first run after start-up measured up to 2.5x slower)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-second-trunk', action='store_true', help='skip the gga_kitti_config.py (sparse trunk) leg')
    ap.add_argument('--second-batch', type=int, default=8, help='frames per GPU of the second_trunk leg')
    ap.add_argument('--no-pgd', action='store_true', help='skip the gga_pdg.py (camera-only retraining) leg')
    ap.add_argument('--pgd-batch', type=int, default=12, help='images per GPU of the pgd leg (samples_per_gpu of the config)')
    ap.add_argument('--no-fcaf3d', action='store_true', help='skip the fcaf3d (SUN RGB-D-shaped scenes) leg')
    ap.add_argument('--fcaf3d-batch', type=int, default=8, help='scenes per GPU of the fcaf3d leg (samples_per_gpu of the config)')
    ap.add_argument('--no-loader-fed', action='store_true', help='skip the dataset-fed leg (on-disk synthetic KITTI tree -> train_detector)')
    ap.add_argument('--loader-frames', type=int, default=2048, help='frames of the synthetic on-disk tree of the loader_fed leg')
    ap.add_argument('--loader-workers', default='4,8', help='workers_per_gpu values the loader_fed leg is run with')
    ap.add_argument('--no-inference', action='store_true', help='skip the inference / pseudo-label legs')
    ap.add_argument('--inference-frames', type=int, default=1024, help='timed frames of each inference pass (after 64 warm-up frames of the same loader)')
    ap.add_argument('--no-planes3', action='store_true',
                    help='skip the `planes3` legs (main config and second_trunk re-timed on three bf16 planes / six products)')
    return ap.parse_args(argv)


def visible_gpus():
    """Number of GPUs this process tree may use, WITHOUT loading a GPU runtime: the visibility variables if set, else the
    KFD topology (nodes with SIMDs are GPUs; CPU nodes have simd_count 0)."""
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(',') if x.strip() != ''])
    n = 0
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get('simd_count', '0')) > 0
    except OSError:
        pass
    return n


def launch_ranks(args):
    """Parent of an N-rank run: nothing in this process touches HIP / HSA (devices are counted from the environment or
    sysfs, torch is not imported). Children are ordinary subprocesses, never an exec."""
    n_dev = visible_gpus()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if n_dev < args.gpus and 'GGA_DIST_BACKEND' not in env:
        env['GGA_DIST_BACKEND'] = 'gloo'      # ranks share devices: RCCL needs one device per rank
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


def damp_head_init(model, scale):
    """Random-init only: Kaiming(fan_out) on the 1-3 channel output convs gives log-dimensions of
    std ~5 (boxes of e^10 m and worse) on noise inputs, i.e. inf/NaN losses that say nothing about
    speed. Scale those output convs so the synthetic run stays finite; architecture, shapes and
    every kernel launched are unchanged."""
    import torch
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(scale)


def run_workload(config, batch, steps, warmup, args, rank, world, device, sites=(), planes=None):
    """Build the model of `config`, step it `warmup` + `steps` times on two resident batches.
    `sites`: (site, cap, key) timing sessions armed for the timed steps (rank 0). `planes`: arithmetic of the matrix
    kernels the Runner selects (config key `gga_dense_planes`; None = the Runner's default, two fp16 planes under its guard).
    -> dict(dt, loss, timings {site: [ms]}, model, batches, cfg)."""
    import torch
    import torch.distributed as dist
    from gga_amd import Config, build_model, synthetic, _lib
    from gga_amd.train import Runner, setup_multi_processes

    cfg = Config.fromfile(config)
    if planes is not None:
        cfg['gga_dense_planes'] = planes
    setup_multi_processes(cfg)          # as tools/train.py:127 does before it builds the model
    channels_last = not args.nchw
    if channels_last and cfg.model.pts_middle_encoder.type in ('PointPillarsScatter', 'SparseEncoder'):
        cfg.model.pts_middle_encoder['channels_last'] = True
    torch.manual_seed(0)
    model = build_model(cfg.model).to(device)
    damp_head_init(model, args.head_init_scale)
    if channels_last:
        from gga_amd.cnn import to_channels_last
        model = to_channels_last(model)
    model.train()
    runner = Runner(model, cfg, max_iters=max(1000, steps + warmup), distributed=world > 1, device=device)

    pc_range = tuple(cfg.model.pts_voxel_layer.point_cloud_range)
    batches = []
    for i in range(2):      # two distinct batches per rank, point clouds resident in HBM
        b = synthetic.make_batch(batch, start=i * batch, rank=rank, pc_range=pc_range)
        b['points'] = [p.to(device) for p in b['points']]
        batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
    torch.cuda.synchronize()
    runner.inputs_ready(*batches)       # resident: the prefetch of a batch need not wait for the main stream's queue

    for i in range(warmup):
        runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    for site, cap, key in sites:      # HIP-event pairs around the kernels, no synchronisation
        _lib.timing_begin(site, cap, key)
    # the product loop's own garbage-collector policy (Runner.run does the same after its first iterations): everything
    # built so far moves to the permanent generation, the collector stays ON in the timed region
    runner.freeze_gc()
    trace = [] if os.environ.get('GGA_BENCH_STEP_TIMES') else None      # diagnosis: host timestamp after every step's queueing
    t0 = time.perf_counter()
    for i in range(steps):      # next_data: the point-only front of the next step is prefetched on a side stream (sparse trunk)
        # (the two batches keep alternating across the warm-up / timed boundary: the batch the last warm-up step prefetched is the
        # one the first timed step consumes, as in a loader-fed loop - with the index restarting at 0 the first timed step of an odd
        # warm-up found the other batch prepared, threw it away and ran its own front inline: +13 ms on the sparse step, +5 on the next)
        out = runner.step(batches[(warmup + i) % 2], next_data=batches[(warmup + i + 1) % 2])
        if trace is not None:
            trace.append(time.perf_counter())
    torch.cuda.synchronize()
    if trace:
        print('host ms between steps (%s): ' % os.path.basename(config) + ' '.join('%.1f' % ((b - a) * 1e3) for a, b in zip([t0] + trace, trace)) +
              '  | drain %.1f' % ((time.perf_counter() - trace[-1]) * 1e3), file=sys.stderr)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timings = {site: _lib.timing_collect(site, cap) for site, cap, _ in sites}
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss = float(out['loss'].detach()) if torch.is_tensor(out['loss']) else float(out['loss'])
    assert loss == loss, f'loss is NaN ({config})'
    return dict(dt=dt, loss=loss, timings=timings, model=model, batches=batches, cfg=cfg, runner=runner)


def run_mono_workload(batch, steps, warmup, args, rank, world, device):
    """configs/gga/gga_pdg.py (BASELINE config #5: PGD retrained on the GGA pseudo labels): FCOSMono3D = ResNet-101
    + FPN + PGDHead (DCNv2 towers), full train step (fwd + bwd + clip + SGD) on synthetic KITTI-mono3d batches
    (1242 x 375 images padded to 1248 x 384, resident in HBM). -> dict(dt, loss)."""
    import torch
    import torch.distributed as dist
    from gga_amd import Config, build_model, synthetic
    from gga_amd.cnn import to_channels_last
    from gga_amd.train import Runner, setup_multi_processes
    cfg = Config.fromfile(PGD_CONFIG)
    setup_multi_processes(cfg)
    torch.manual_seed(0)
    model = build_model(cfg.model).to(device)
    import warnings
    with warnings.catch_warnings():        # the model-zoo checkpoint of the backbone cannot be fetched here: random init, said once below
        warnings.simplefilter('ignore')
        model.init_weights()               # as tools/train.py:222 does after build_model
    synthetic.damp_random_backbone(model)      # stand-in for the unavailable checkpoint's statistics (see its docstring)
    if not args.nchw:
        model = to_channels_last(model)
    model.train()
    runner = Runner(model, cfg, max_iters=max(1000, steps + warmup), distributed=world > 1, device=device, iters_per_epoch=1000)
    batches = []
    for i in range(2):
        b = synthetic.make_mono_batch(batch, start=i * batch, rank=rank, device=device)
        if not args.nchw:
            b['img'] = b['img'].contiguous(memory_format=torch.channels_last)
        batches.append({k: b[k] for k in synthetic.MONO_BATCH_KEYS})
    torch.cuda.synchronize()
    for i in range(warmup):
        runner.step(batches[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    runner.freeze_gc()
    t0 = time.perf_counter()
    for i in range(steps):
        out = runner.step(batches[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    loss = float(out['loss'].detach())
    assert loss == loss, 'loss is NaN (gga_pdg.py)'
    return dict(dt=float(t.item()), loss=loss, runner=runner, batches=batches)


def run_indoor_workload(batch, steps, warmup, args, rank, world, device):
    """configs/fcaf3d/fcaf3d_8x2_sunrgbd-3d-10class.py (BASELINE config #4: FCAF3D on SUN RGB-D-shaped scenes, 50 k points, 10
    classes): MinkResNet-34 + the sparse FPN / head of FCAF3DHead on the gather-GEMM kernels, full train step (fwd + bwd +
    clip + AdamW). There is no GGA head for this trunk in the reference: stock FCAF3D. -> dict(dt, loss)."""
    import torch
    import torch.distributed as dist
    from gga_amd import Config, build_model, synthetic
    from gga_amd.train import Runner, setup_multi_processes
    cfg = Config.fromfile(FCAF3D_CONFIG)
    setup_multi_processes(cfg)
    torch.manual_seed(0)
    model = build_model(cfg.model).to(device).train()
    runner = Runner(model, cfg, max_iters=max(1000, steps + warmup), distributed=world > 1, device=device, iters_per_epoch=1000)
    batches = []
    for i in range(2):
        b = synthetic.make_indoor_batch(batch, start=i * batch, rank=rank, device=device, n_points=50000)
        batches.append({k: b[k] for k in synthetic.INDOOR_BATCH_KEYS})
    torch.cuda.synchronize()
    for i in range(warmup):
        runner.step(batches[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    runner.freeze_gc()
    t0 = time.perf_counter()
    for i in range(steps):
        out = runner.step(batches[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    loss = float(out['loss'].detach())
    assert loss == loss, 'loss is NaN (fcaf3d)'
    return dict(dt=float(t.item()), loss=loss, runner=runner)


def pipeline_stage_ms(dataset, frames=48):
    """ms per frame of every stage of the train pipeline, run in THIS process on the first `frames` frames."""
    import numpy as np
    import torch
    inner = getattr(dataset, 'dataset', dataset)
    np.random.seed(0), torch.manual_seed(0)
    acc, objects, points = {}, 0, 0
    for i in range(frames):
        t = time.perf_counter()
        sample = inner.get_data_info(i)
        inner.pre_pipeline(sample)
        acc['get_data_info'] = acc.get('get_data_info', 0.0) + time.perf_counter() - t
        for tr in inner.pipeline.transforms:
            t = time.perf_counter()
            sample = tr(sample)
            acc[type(tr).__name__] = acc.get(type(tr).__name__, 0.0) + time.perf_counter() - t
        objects += len(sample['gt_labels_3d'].data)
        points += int(sample['points'].data.shape[0])
    out = {k: round(v / frames * 1e3, 3) for k, v in acc.items()}
    out['total'] = round(sum(acc.values()) / frames * 1e3, 3)
    return out, objects / frames, points / frames


def bench_tree_root(suffix=''):
    """Directory of the synthetic on-disk tree: GGA_BENCH_TREE, else a per-user directory under the temp dir that is only
    reused when this user owns it and nobody else can write to it (its pickles are loaded: ADVICE r05); otherwise a fresh one."""
    import stat
    import tempfile
    root = os.environ.get('GGA_BENCH_TREE')
    if root:
        return root + suffix
    root = os.path.join(tempfile.gettempdir(), f'gga_bench_kitti_{os.getuid()}{suffix}')
    try:
        os.makedirs(root, mode=0o700, exist_ok=True)
        st = os.stat(root)
        if st.st_uid == os.getuid() and not (st.st_mode & (stat.S_IWGRP | stat.S_IWOTH)):
            return root
    except OSError:
        pass
    return tempfile.mkdtemp(prefix='gga_bench_kitti_')


def run_loader_fed(args, device, resident_value):
    """The REAL train job on one GPU (mmdet3d/apis/train.py:180-322 through gga_amd.train.train_detector): a synthetic KITTI tree
    on disk (scans, the GGA info file, the GT database: synthetic.write_kitti_tree) -> KittiDataset_GGA_train with the
    reference's full train_pipeline of configs/gga/gga_kitti_config.py:93-137 (ObjectSample_GGA with database sampling,
    range filters, shuffle, format, collect) in `workers_per_gpu` loader workers -> collate -> upload -> Runner.step, one epoch;
    the first iterations are warm-up (worker start-up, allocator), the rest is timed between two device synchronisations.
    -> dict per workers_per_gpu: frames/s, share of the loop's wall time spent waiting in next(loader), host ms per iteration
    by part; plus the pipeline's ms per frame by stage (this process, one thread)."""
    import tempfile
    import torch
    from gga_amd import Config, build_model, synthetic
    from gga_amd.cnn import to_channels_last
    from gga_amd.loader import build_dataset
    from gga_amd.train import setup_multi_processes, train_detector
    root = bench_tree_root()
    t0 = time.perf_counter()
    info_path, db_path = synthetic.write_kitti_tree(root, args.loader_frames, pc_range=synthetic.RANGE_PP)
    tree_s = time.perf_counter() - t0
    out = {'workload': f'train_detector on {os.path.relpath(PP_CONFIG, REPO)}: {args.loader_frames} synthetic KITTI frames of 20 000 points on disk '
                       f'(+ GT database, 300 objects per class), the reference\'s full train_pipeline incl. ObjectSample_GGA database sampling, '
                       f'samples_per_gpu {args.batch}, one epoch; same model / arithmetic / step as the headline',
           'tree_seconds': round(tree_s, 1), 'resident_batch_frames_per_s': resident_value}
    warm = 8
    for workers in [int(w) for w in args.loader_workers.split(',') if w]:
        cfg = Config.fromfile(PP_CONFIG)
        d = cfg.data['train']
        d['dataset'].update(data_root=root + '/', ann_file=info_path)
        for t in d['dataset']['pipeline']:
            if t['type'] == 'ObjectSample_GGA':
                t['db_sampler'].update(data_root=root + '/', info_path=db_path)
            if 'point_cloud_range' in t:          # (the base file's pipeline carries the sparse config's range)
                t['point_cloud_range'] = list(synthetic.RANGE_PP)
        cfg.data.update(samples_per_gpu=args.batch, workers_per_gpu=workers)
        cfg.runner = dict(type='EpochBasedRunner', max_epochs=1)
        cfg.checkpoint_config, cfg.work_dir, cfg.seed = None, None, 0
        setup_multi_processes(cfg)
        if not args.nchw:
            cfg.model.pts_middle_encoder['channels_last'] = True
        torch.manual_seed(0)
        model = build_model(cfg.model).to(device)
        damp_head_init(model, args.head_init_scale)
        if not args.nchw:
            model = to_channels_last(model)
        model.train()
        dataset = build_dataset(d)
        if 'pipeline_ms_per_frame' not in out:
            stages, objs, pts = pipeline_stage_ms(dataset)
            out['pipeline_ms_per_frame'] = stages
            out['objects_per_frame_after_sampling'] = round(objs, 1)
            out['points_per_frame_after_sampling'] = round(pts)
        mark = {}

        def after_iter(runner, n):
            if n == warm:
                torch.cuda.synchronize()
                mark['t0'] = time.perf_counter()
                for k in runner.loop_seconds:
                    runner.loop_seconds[k] = 0
        runner = train_detector(model, dataset, cfg, distributed=False, device=device, after_iter=after_iter)
        torch.cuda.synchronize()
        dt = time.perf_counter() - mark['t0']
        ls = runner.loop_seconds
        iters = ls['iters']
        assert iters >= 40, f'{iters} timed iterations: raise --loader-frames'
        out[f'workers_{workers}'] = {
            'value': round(iters * args.batch / dt, 3), 'unit': 'frames/s', 'ms_per_step': round(dt / iters * 1e3, 3),
            'timed_steps': iters, 'warmup_steps': warm, 'vs_resident_batches': round(iters * args.batch / dt / resident_value, 3),
            'data_wait_fraction': round(ls['fetch'] / dt, 4),
            'host_ms_per_step': {'next(loader)': round(ls['fetch'] / iters * 1e3, 3), 'unpack + upload': round(ls['inputs'] / iters * 1e3, 3),
                                 'Runner.step (queueing)': round(ls['step'] / iters * 1e3, 3)},
            'matrix_planes': runner.planes}
        del runner, model, dataset
        gc.collect()
        torch.cuda.empty_cache()
    return out


def run_inference(args, device, model_config, pc_range, frames=1024, batch_sizes=(1, 16)):
    """The pseudo-label run of the recipe (reference: tools/generate_pseudo_labels_gga.py:242 /
    tools/test.py -> mmdet3d/apis/test.py single_gpu_test -> MVXTwoStageDetector_GGA.simple_test, centerpoint_head_gga.py:725-934
    get_bboxes / get_task_detections, core/post_processing/box3d_nms.py:231-268; fps probe tools/analysis_tools/benchmark.py:66-91):
    configs/gga/gga_kitti_matching_config.py's test section on the synthetic on-disk tree - LoadPointsFromFile + the test
    pipeline in loader workers, max_voxels = 40000 (eval), trunk, head, decode (top-100 per task), score threshold, BEV
    rotated NMS, bbox3d2result - `frames` frames after the first 64 of the same loader, for `samples_per_gpu` 1 (the reference tool's default)
    and 16; then KittiDataset_GGA_match.evaluate (detections -> camera frame -> image-plane IoU matching against the 2D boxes
    -> the pseudo-label file). Random-init weights with the heat-map bias raised so that the detector reports boxes.
    -> dict per samples_per_gpu: frames/s, ms per frame by stage (a second, synchronised pass)."""
    import copy
    import tempfile
    import torch
    from gga_amd import Config, build_model, synthetic, ops
    from gga_amd.apis import single_gpu_test
    from gga_amd.cnn import to_channels_last
    from gga_amd.loader import build_dataloader, build_dataset
    is_pp = 'pointpillars' in os.path.basename(model_config)
    root = bench_tree_root('' if is_pp else '_second')
    info_path, _ = synthetic.write_kitti_tree(root, max(frames + 64, args.loader_frames if is_pp else frames + 64), pc_range=pc_range)
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_matching_config.py'))
    mcfg = Config.fromfile(model_config)
    test = dict(cfg.data['test'])
    pipe = copy.deepcopy(list(test['pipeline']))
    for t in pipe[1]['transforms']:
        if t['type'] == 'PointsRangeFilter':
            t['point_cloud_range'] = list(pc_range)
    test.update(data_root=root + '/', ann_file=info_path, pipeline=pipe, pcd_limit_range=list(pc_range), test_mode=True)
    model_cfg = mcfg.model
    model_cfg['train_cfg'] = None
    if not args.nchw:
        model_cfg['pts_middle_encoder']['channels_last'] = True
    torch.manual_seed(0)
    model = build_model(model_cfg)
    damp_head_init(model, args.head_init_scale)
    model = model.to(device)
    if not args.nchw:
        model = to_channels_last(model)
    model.CLASSES = ('Pedestrian', 'Cyclist', 'Car')
    model.eval()
    import pickle
    infos = pickle.load(open(info_path, 'rb'))

    def subset(lo, hi, name):           # a dataset of exactly these frames (evaluate wants one result per frame of ITS dataset)
        path = os.path.join(root, f'kitti_infos_{name}.pkl')
        pickle.dump(infos[lo:hi], open(path, 'wb'))
        return build_dataset(dict(test, ann_file=path))
    WARM = 64 if frames >= 256 else 32
    timed_set, stage_set = subset(0, frames + WARM, 'bench_timed'), subset(0, min(64, frames) + WARM, 'bench_stages')
    eval_set = subset(WARM, frames + WARM, 'bench_eval')
    out = {'workload': (f'single_gpu_test on {os.path.relpath(model_config, REPO)} + the test section of configs/gga/gga_kitti_matching_config.py: '
                        f'{frames} synthetic KITTI frames of 20 000 points from disk through the test pipeline (8 loader workers), voxelize at max_voxels 40000, '
                        f'trunk, head, decode top-100 per task, score threshold 0.1, BEV rotated NMS, then KittiDataset_GGA_match.evaluate'),
           'frames': frames, 'matrix_planes': 2, 'weights': 'random init (heat-map bias -2.19 = a score of 0.10 before the trunk\'s noise: about half of every task\'s 100 candidates pass the 0.1 threshold)'}

    stages = {}

    def timed(name, fn):
        def wrapper(*a, **kw):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn(*a, **kw)
            torch.cuda.synchronize()
            stages[name] = stages.get(name, 0.0) + time.perf_counter() - t
            return r
        return wrapper
    head = model.pts_bbox_head
    for spg in batch_sizes:
        mk = lambda ds: build_dataloader(ds, samples_per_gpu=spg, workers_per_gpu=8, dist=False, shuffle=False)
        # one loader over WARM + `frames` frames; the clock starts when the first WARM results are in (loader workers started,
        # allocator warm - the reference's fps probe skips its first iterations the same way, benchmark.py:66-91)
        mark = {}

        def progress(n):
            if 't0' not in mark and n >= WARM:
                torch.cuda.synchronize()
                mark['t0'], mark['n0'] = time.perf_counter(), n
        results = single_gpu_test(model, mk(timed_set), device, progress=progress, planes=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - mark['t0']
        results = results[mark['n0']:]
        assert mark['n0'] == WARM, mark
        assert len(results) == frames
        dets = sum(len(r['pts_bbox']['scores_3d']) for r in results) / frames
        # second pass over 64 frames with a device synchronisation around every stage (slower than the pass above: the stages add up
        # to more than its time per frame)
        stages.clear()
        real = dict(voxelize=model.voxelize, enc=model.pts_voxel_encoder.forward, mid=model.pts_middle_encoder.forward,
                    bb=model.pts_backbone.forward, neck=model.pts_neck.forward, head=head.forward, decode=head.bbox_coder.decode,
                    nms=ops.nms_bev, get=head.get_bboxes)
        model.voxelize = timed('voxelize', real['voxelize'])
        model.pts_voxel_encoder.forward = timed('trunk', real['enc'])
        model.pts_middle_encoder.forward = timed('trunk', real['mid'])
        model.pts_backbone.forward = timed('trunk', real['bb'])
        model.pts_neck.forward = timed('trunk', real['neck'])
        head.forward = timed('head', real['head'])
        head.get_bboxes = timed('get_bboxes (decode + threshold + nms + merge)', real['get'])
        head.bbox_coder.decode = timed('  of which decode', real['decode'])
        ops.nms_bev = timed('  of which nms_bev', real['nms'])
        try:
            n_st = len(stage_set) - WARM
            mark2 = {}

            def progress2(n):
                if 't0' not in mark2 and n >= WARM:
                    torch.cuda.synchronize()
                    mark2['t0'] = time.perf_counter()
                    stages.clear()
            single_gpu_test(model, mk(stage_set), device, progress=progress2, planes=2)
            torch.cuda.synchronize()
            st_total = time.perf_counter() - mark2['t0']
        finally:
            model.voxelize = real['voxelize']
            model.pts_voxel_encoder.forward, model.pts_middle_encoder.forward = real['enc'], real['mid']
            model.pts_backbone.forward, model.pts_neck.forward = real['bb'], real['neck']
            head.forward, head.get_bboxes, head.bbox_coder.decode, ops.nms_bev = real['head'], real['get'], real['decode'], real['nms']
        per = {k: round(v / n_st * 1e3, 3) for k, v in stages.items()}
        per['loader wait + upload + result hand-over (remainder of the synchronised pass)'] = round(
            (st_total - sum(v for k, v in stages.items() if not k.startswith('  '))) / n_st * 1e3, 3)
        out[f'samples_per_gpu_{spg}'] = {'value': round(frames / dt, 2), 'unit': 'frames/s', 'ms_per_frame': round(dt / frames * 1e3, 3),
                                         'detections_per_frame': round(dets, 1), 'stage_ms_per_frame_synchronised': per}
        last = results
    # the matching step (host + gga_image_box_match): detections -> pseudo labels of the same frames
    t0 = time.perf_counter()
    res = eval_set.evaluate(last, metric=['mAP'], device=str(device),
                             pseudo_label_file=os.path.join(tempfile.gettempdir(), f'gga_bench_pseudo_{os.getuid()}.pkl'))
    torch.cuda.synchronize()
    out['match_ms_per_frame'] = round((time.perf_counter() - t0) / frames * 1e3, 3)
    out['pseudo_labels'] = {k: v for k, v in (res or {}).items() if k.startswith('pseudo_labels/')}
    del model
    gc.collect()
    torch.cuda.empty_cache()
    return out


def scatter_roofline(model, batches, step_ms):
    """`roofline` of the pillar-scatter canvas kernel: algorithmic bytes over
    its mean duration INSIDE the timed steps (`step_ms`: one HIP-event pair per step around the
    kernel, on the stream it is launched on)."""
    import torch
    vl, me = model.pts_voxel_layer, model.pts_middle_encoder
    B = len(batches[0]['points'])
    ch = me.in_channels
    with torch.no_grad():
        coors = [vl.forward_batch(b['points'])[2] for b in batches]
    m = sum(c.shape[0] for c in coors) / len(coors)
    algo = m * ch * 4 + m * 16 + B * ch * me.ny * me.nx * 4
    ms = sum(step_ms) / len(step_ms)
    gbs = algo / (ms * 1e-3) / 1e9
    knames = (['scatter_fill_kernel', 'scatter_rows_nhwc_kernel'] if me.channels_last else ['scatter_canvas_nchw_v2_kernel'])
    # HBM bytes per launch: NOT measured in this run. Taken from the PMC passes kept under profiles/
    # (FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE exact) and only quoted when the shape is the profiled one.
    traffic, src = None, None
    for name in ('r06_scatter_pmc.json', 'r05_scatter_pmc.json', 'r04_scatter_pmc.json', 'r02_scatter_pmc.json', 'r01_scatter_pmc.json'):
        try:
            pmc = json.load(open(os.path.join(REPO, 'profiles', name)))
            if (B, ch, me.ny, me.nx) == (16, 64, 496, 432) and abs(m - 256000) < 2000:
                traffic = sum(pmc['kernels'][k]['hbm_bytes_corrected'] for k in knames)
                src = f'profiles/{name} (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not this run)'
                break
        except (OSError, KeyError, ValueError):
            continue
    return {'bound': 'hbm', 'kernel': '+'.join(knames), 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': src, 'algorithmic_bytes': int(algo),
            'kernel_ms': round(ms, 4), 'launches_timed': len(step_ms), 'timed': 'in-step, HIP events on the launch stream',
            'pillars': int(m)}


def sparse_roofline(run, args, steps=4):
    """`roofline` of the shipped config's dominant kernel - the 128 -> 128 submanifold convolutions of the sparse trunk's
    last two stages (sp_conv_x9_kernel<4, ...>: forward and backward-data) - from a few extra steps after the timed region:
    HIP events around exactly those launches, their sizes read off the calls. The bound is HBM: the kernel is a gather.
    ALGORITHMIC bytes per launch = every feature row once in and once out + the rule book + the weight planes
    (rows * (Cin + Cout) * 4 + kvol * rows * 4 + kvol * Cin * Cout * 4); `traffic` = the HBM-side bytes per launch the PMC passes
    kept under profiles/ counted for the same kernel (not this run) - far above the algorithmic figure: a row is fetched once
    per (output row, offset) pair that uses it (14.5 times on average), which is what there is to fix."""
    from gga_amd import _lib
    L = _lib.lib()
    seen = []
    real, real_halo = L.gga_sparse_conv_apply_bn_bwd, L.gga_sparse_conv_apply_halo

    def spy(*a):
        if a[7] == 128 and a[8] == 128 and a[6] == 27:
            seen.append((int(a[5]), 'sp_conv_x9_kernel<4,true,2>'))
        return real(*a)

    def spy_halo(*a):                       # (x, weights, tile rows, counts, capacity, halo rows, local map, n_rows, n_tiles, kvol, cin, cout, ...)
        if a[10] == 128 and a[11] == 128 and a[9] == 27:
            seen.append((int(a[7]), 'sp_conv_halo_kernel<4>'))
        return real_halo(*a)
    L.gga_sparse_conv_apply_bn_bwd, L.gga_sparse_conv_apply_halo = spy, spy_halo
    from gga_amd import sparse as _sparse
    tilings = []                            # the halo tilings built during these steps (for the number of real pairs)
    halo_init = _sparse._Halo.__init__

    def halo_spy(self, *a, **kw):
        halo_init(self, *a, **kw)
        tilings.append(self)
    _sparse._Halo.__init__ = halo_spy
    ms = []
    try:
        # the halo form's launches carry the key (cin, cout, kvol), the default kernel's (cin, cout, 0) - the latter shared by
        # every 128 -> 128 gather-GEMM of the step, whatever its size: taken only when no launch took the halo form
        for hw in (27, 0):
            del seen[:]
            _lib.timing_begin(_lib.TIME_SPARSE_CONV, 64 * steps, _lib.timing_conv_key(128, 128, hw))
            for i in range(steps):       # (every rank takes these steps: they contain the gradient all-reduce)
                run['runner'].step(run['batches'][i % 2], next_data=run['batches'][(i + 1) % 2])
            ms = _lib.timing_collect(_lib.TIME_SPARSE_CONV, 64 * steps)
            if ms:
                break
    finally:
        L.gga_sparse_conv_apply_bn_bwd, L.gga_sparse_conv_apply_halo = real, real_halo
        _sparse._Halo.__init__ = halo_init
    halo = bool(ms) and any(k == 'sp_conv_halo_kernel<4>' for _, k in seen)
    big = [(n_, k) for n_, k in seen if (k == 'sp_conv_halo_kernel<4>') == halo]
    if not ms or not big:
        return None
    names = sorted(set(k for _, k in big))
    rows = sum(n for n, _ in big) / len(big)
    algo = rows * (128 + 128) * 4 + 27 * rows * 4 + 27 * 128 * 128 * 4
    avg = sum(ms) / len(ms)
    traffic, src = None, None
    for name in ('r06_second_pmc.json', 'r05_second_pmc.json', 'r04_second_pmc.json'):      # the newest PMC pass that holds this kernel
        try:
            pmc = json.load(open(os.path.join(REPO, 'profiles', name)))
            ks = [k for k in pmc['kernels'] if k['kernel'].startswith('sp_conv_halo_kernel<4' if halo else 'sp_conv_x9_kernel<4') and 'hbm_bytes_per_launch' in k
                  and k['avg_us'] > 500]
            if ks and args.second_batch == 8:
                traffic = int(sum(k['hbm_bytes_per_launch'] * k['launches_per_pass'] for k in ks) / sum(k['launches_per_pass'] for k in ks))
                src = f'profiles/{name} (separate --pmc FETCH_SIZE / WRITE_SIZE passes over tools_dev/pmc_target_second.py, not this run)'
                break
        except (OSError, KeyError, ValueError):
            continue
    gbs = algo / (avg * 1e-3) / 1e9
    mfma = None
    if halo and tilings:
        # matrix work of one launch: REAL (row, offset) pairs x 2 Cin Cout x three partial products (what VERDICT r02 asked to
        # see), and what the halo form issues - all 27 offsets of every 64-row block
        t = tilings[-1]
        pairs = int((t.local_map != -1).sum())
        useful = pairs * 2.0 * 128 * 128 * 3 / (avg * 1e-3) / 1e12
        issued = t.n_tiles * 256 * 27 * 2.0 * 128 * 128 * 3 / (avg * 1e-3) / 1e12
        mfma = {'bound': 'mfma', 'achieved': round(useful, 1), 'peak': BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(useful / BF16_PEAK_TFLOPS, 4),
                'pairs': pairs, 'pairs_per_row': round(pairs / rows, 2), 'partial_products': 3,
                'issued_tflops': round(issued, 1), 'issued_frac': round(issued / BF16_PEAK_TFLOPS, 4),
                'note': 'achieved counts the real (row, offset) pairs; issued counts the 27 offsets the halo form multiplies for every row'}
    kernel = ' + '.join(names) + ' (SubMConv3d 128 -> 128, 27 offsets, forward + backward-data)'
    common = {'kernel_ms': round(avg, 4), 'launches_timed': len(ms), 'launches_per_step': len(ms) / steps,
              'timed': 'in-step, HIP events on the launch stream', 'rows': int(rows)}
    hbm = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
           'traffic': traffic, 'traffic_source': src, 'traffic_GBps': round(traffic / (avg * 1e-3) / 1e9, 1) if traffic else None,
           'algorithmic_bytes': int(algo)}
    if mfma:
        # the halo form stages every row once per chunk and is bounded by its matrix work (PMC: 0.47-0.48 of the pipe's cycles at
        # 2.0-2.1 GHz, 1.56 x the algorithmic bytes at 0.9 TB/s: EXPERIMENTS.md 6c / 6d) - that is the roofline of the line; the byte
        # side stays beside it
        return dict(mfma, kernel=kernel, traffic=traffic, hbm_roofline=hbm, **common)
    return dict(hbm, kernel=kernel, mfma_roofline=None, **common)


def mfma_roofline(kernel, flops, ms_list, launches_per_step, ms_per_step, planes=2):
    """16-bit matrix work issued = products x the fp32 FLOPs (three partial products on two fp16 planes, six on three
    bf16 planes); peak = dense bf16 / f16 MFMA (the same rate)."""
    products = 3 if planes == 2 else 6
    avg = sum(ms_list) / len(ms_list)
    tf = products * flops / (avg * 1e-3) / 1e12
    return {'bound': 'mfma', 'kernel': kernel, 'achieved': round(tf, 1), 'peak': BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(tf / BF16_PEAK_TFLOPS, 4), 'partial_products': products,
            'fp32_equivalent_tflops': round(flops / (avg * 1e-3) / 1e12, 1),
            'kernel_ms': round(avg, 4), 'launches_timed': len(ms_list), 'launches_per_step': launches_per_step,
            'timed': 'in-step, HIP events on the launch stream',
            # mean kernel time x launches per step (independent of how many launches the session sampled)
            'share_of_step': round(avg * launches_per_step / ms_per_step, 3)}


def _median_time(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_pieces(cfg, frames=16):
    """the oracle's pieces timed on the host (3 warm-ups + 10 timed, median) on the
    same synthetic frames: (i) voxelize, (ii) pillar scatter, (iii) target generation,
    (iv) 3D->2D projection + BPL/SRL geometry, (v) point-to-box alignment. ms per 16-frame batch."""
    import numpy as np
    import torch
    from gga_amd import build_model, synthetic
    from oracle import oracle as O
    from oracle import torch_ref as R
    torch.manual_seed(0)
    model = build_model(cfg.model)
    head, vl, me = model.pts_bbox_head, model.pts_voxel_layer, model.pts_middle_encoder
    tc = head.train_cfg
    batch = synthetic.make_batch(frames, start=901, pc_range=synthetic.RANGE_PP)
    pts = [np.ascontiguousarray(p.numpy(), np.float32) for p in batch['points']]
    out = {}
    out['voxelize_ms'] = _median_time(lambda: O.voxelize_batch(pts, vl.voxel_size, vl.point_cloud_range, vl.max_num_points,
                                                               vl.max_voxels[0])) * 1e3
    v, n, c = O.voxelize_batch(pts, vl.voxel_size, vl.point_cloud_range, vl.max_num_points, vl.max_voxels[0])
    feats, coors = torch.randn(len(c), 64), torch.from_numpy(c)
    out['scatter_ms'] = _median_time(lambda: R.scatter(feats, coors, frames, me.ny, me.nx)) * 1e3
    as_np = lambda xs: [a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a) for a in xs]
    srl = O.draw_srl(frames, len(head.task_heads))
    tg_args = (as_np(batch['gt_labels_3d']), as_np(batch['GGA_boxes_img']), as_np(batch['GGA_lidar2img']),
               as_np(batch['GGA_init_pseudo_labels']), as_np(batch['GGA_bdry_masks']),
               [as_np(f) for f in batch['GGA_in_box_points']], [m['lidar2img'] for m in batch['img_metas']], tc, srl)
    out['targets_ms'] = _median_time(lambda: O.get_targets(*tg_args, n_tasks=len(head.task_heads))) * 1e3
    tg = O.get_targets(*tg_args, n_tasks=len(head.task_heads))
    K = tg['ind'][0].shape[1]
    pred = [torch.randn(frames, K, 8) * 0.1 for _ in range(len(head.task_heads))]

    def proj():
        for t in range(len(head.task_heads)):
            R.box_geometry(pred[t], torch.from_numpy(tg['ind'][t]), torch.from_numpy(tg['lidar2img'][t]), tc)
    out['projection_ms'] = _median_time(proj) * 1e3
    bevs = [R.box_geometry(pred[t], torch.from_numpy(tg['ind'][t]), torch.from_numpy(tg['lidar2img'][t]), tc)[3]
            for t in range(len(head.task_heads))]

    def pal():
        for t in range(len(head.task_heads)):
            R.pal_distances(tg['ibp'][t], bevs[t])
    out['pal_ms'] = _median_time(pal, warm=1, reps=5) * 1e3
    return {k: round(v, 2) for k, v in out.items()}


def bench_size_gpu_losses(run):
    """Whole-step parity AT THE BENCH SIZE, GPU side: the 18 losses of one forward pass of the timed
    model - its weights as the timed steps left them - on `batches[0]` of the timed loop (16 frames x 20 000 points), for
    both arithmetic forms, with SRL draws that are handed to the CPU step (`cpu_baseline`) together with the weights and the
    batch. -> dict(state, batch, srl, gpu={planes: {key: loss}})."""
    import torch
    from gga_amd import dense_conv
    model, batch = run['model'], run['batches'][0]
    B = len(batch['points'])
    srl = model.pts_bbox_head.draw_srl(B)
    gpu = {}
    was = dense_conv.PLANES
    try:
        for planes in (2, 3):
            dense_conv.PLANES = planes
            dense_conv.AMAX_POOL.next_generation()
            feats = model.extract_feat(batch['points'], None, batch['img_metas'])[1]
            outs = model.pts_bbox_head(feats)
            losses = model.pts_bbox_head.loss(batch['gt_bboxes_3d'], batch['gt_labels_3d'], outs, batch['GGA_boxes_img'],
                                              batch['GGA_lidar2img'], batch['GGA_init_pseudo_labels'], batch['GGA_bdry_masks'],
                                              batch['GGA_in_box_points'], batch['img_metas'], srl=srl)
            gpu[planes] = {k: float(v.detach()) for k, v in losses.items()}
            del feats, outs, losses
    finally:
        dense_conv.PLANES = was
    torch.cuda.synchronize()
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cpu_batch = dict(batch, points=[p.detach().cpu() for p in batch['points']])
    return dict(state=state, batch=cpu_batch, srl=srl, gpu=gpu, steps_taken=int(run['runner'].iter))


def cpu_baseline(cfg, frames=16, parity=None):
    """The oracle's CPU restatement of the same train step (C voxelizer + torch fp32 on the
    host cores), one timed step on `frames` frames after a 1-frame warm-up, and its pieces.
    `parity` (bench_size_gpu_losses): the timed step then runs on the GPU legs' own weights and batch with the same SRL
    draws, and its losses are the fp32 side of `parity_at_bench_size` (the float64 side: one more forward pass)."""
    import torch
    from gga_amd import build_model, synthetic
    from oracle import torch_ref as R
    # torch's CPU convolutions stop scaling (and oversubscribe badly) far below the 256 hardware
    # threads of the GPU host: use at most 32 threads and report that number as `cores`
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = build_model(cfg.model)
    damp_head_init(model, 0.05)
    if parity is not None:
        model.load_state_dict(parity['state'])
    model.train()
    warm = synthetic.make_batch(1, start=900, n_points=2000, pc_range=synthetic.RANGE_PP, n_obj_range=(2, 3))
    R.reference_train_step(model, warm)      # first-touch / thread-pool warm-up
    model.zero_grad()
    if parity is not None:
        model.load_state_dict(parity['state'])       # (the warm-up step moved the BatchNorm running statistics: put them back)
        batch, frames = parity['batch'], len(parity['batch']['points'])
    else:
        batch = synthetic.make_batch(frames, start=901, pc_range=synthetic.RANGE_PP)
    t0 = time.perf_counter()
    cpu_losses, _ = R.reference_train_step(model, batch, srl=parity['srl'] if parity is not None else None)
    dt = time.perf_counter() - t0
    parity_out = None
    if parity is not None:
        # the same step in float64 (forward only): the value every fp32 path approximates, and the fp32 CPU step's own distance
        # from it (`fp32_floor`); deviations as in tests/test_model_gpu.py: |a - b| / max(|b|, 1)
        import copy
        t1 = time.perf_counter()
        with torch.no_grad():
            l64, _ = R.reference_train_step(copy.deepcopy(model).double(), batch, srl=parity['srl'], backward=False)
        f64_s = time.perf_counter() - t1
        l32 = {k: float(v) for k, v in cpu_losses.items()}
        l64 = {k: float(v) for k, v in l64.items()}
        rel = lambda a, b: abs(a - b) / max(abs(b), 1.0)
        floor_key = max(l64, key=lambda k: rel(l32[k], l64[k]))
        parity_out = {'what': (f'the 18 losses of ONE forward pass on the batch and weights of the timed GPU steps ({frames} frames x 20 000 points; weights '
                               f'after {parity["steps_taken"]} optimizer steps; same SRL draws): GPU legs against the fp32 CPU step timed here '
                               f'and against the same step in float64; deviation = |a - b| / max(|b|, 1); bound 1e-4 (north_star)'),
                      'fp32_floor': rel(l32[floor_key], l64[floor_key]), 'fp32_floor_key': floor_key, 'float64_forward_s': round(f64_s, 1)}
        for planes, got in parity['gpu'].items():
            k64 = max(l64, key=lambda k: rel(got[k], l64[k]))
            k32 = max(l32, key=lambda k: rel(got[k], l32[k]))
            rms = (sum(rel(got[k], l64[k]) ** 2 for k in l64) / len(l64)) ** 0.5
            parity_out[f'planes{planes}'] = {'planes': planes, 'worst_rel': rel(got[k64], l64[k64]), 'key': k64,
                                             'worst_rel_vs_fp32_cpu': rel(got[k32], l32[k32]), 'key_vs_fp32_cpu': k32,
                                             'rms_rel_over_keys': rms, 'within_1e-4': bool(rel(got[k64], l64[k64]) <= 1e-4 and rel(got[k32], l32[k32]) <= 1e-4)}
        parity_out['fp32_floor_rms_over_keys'] = (sum(rel(l32[k], l64[k]) ** 2 for k in l64) / len(l64)) ** 0.5
    cpu_model = ''
    try:
        cpu_model = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
    except (OSError, IndexError):
        pass
    # BASELINE.md 3's own protocol - 3 warm-up + 10 timed iterations, median - run to its full counts on a 2-frame sample of
    # the same frames (13 steps of the 16-frame sample would take ~7 minutes)
    small = synthetic.make_batch(2, start=901, pc_range=synthetic.RANGE_PP)
    times, warm_done = [], 0
    for i in range(13):
        model.zero_grad()
        t1 = time.perf_counter()
        R.reference_train_step(model, small)
        d1 = time.perf_counter() - t1
        if i < 3:
            warm_done += 1
        else:
            times.append(d1)
    times.sort()
    med = times[len(times) // 2]
    out = {'value': round(frames / dt, 4), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
           'sample': (f'1 train step (fwd+bwd, no optimizer) on {frames} synthetic frames, {dt:.1f} s'
                      + (' - the batch and the weights of the timed GPU steps (see parity_at_bench_size)' if parity is not None else '')),
           'protocol_sample': {'value': round(2 / med, 4), 'unit': 'frames/s', 'frames': 2, 'warmup_iterations': warm_done,
                               'timed_iterations': len(times), 'median_s': round(med, 3), 'threads': cores,
                               'note': 'BASELINE.md 3: 3 warm-up + 10 timed iterations, median, on a 2-frame sample; threads = `cores`, not '
                                       'os.cpu_count(): torch\'s CPU convolutions are slower with all hardware threads of this host than with 32 - measured round 6 on '
                                       'the 256-thread EPYC 9575F host: this 2-frame step takes 151 s with torch.set_num_threads(256) against 2.8 s with 32 '
                                       '(0.013 against 0.72 frames/s; tools: bench.py history, EXPERIMENTS.md 6f)'},
           'protocol': ('`value`: one small warm-up step, ONE timed step of the full 16-frame sample on 32 threads (a step takes ~30 s: 13 of '
                        'them would break the bound on the bench run). `protocol_sample`: BASELINE.md 3\'s 3 + 10 protocol on 2 frames. '
                        'The pieces below follow the protocol too (median of 10 after 3)'),
           'host_cpu': cpu_model, 'host_threads': os.cpu_count(),
           'pieces_ms_per_16_frames': cpu_pieces(cfg, 16)}
    return out, parity_out


def main():
    if os.environ.get('GGA_BENCH_HANG_DUMP') and 'RANK' in os.environ:       # debugging aid (ranks of an N-rank run): every thread's Python stack to stderr after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['GGA_BENCH_HANG_DUMP']), exit=True)
    args = parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        launch_ranks(args)

    import torch
    import torch.distributed as dist
    import gga_amd  # noqa: F401
    from gga_amd import _lib
    from gga_amd.train import init_dist

    rank, world, local_rank = init_dist()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (the product has no CPU path)'
    device = torch.device('cuda', local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    backend = dist.get_backend() if world > 1 else 'none'
    if backend == 'nccl':
        backend = 'nccl (RCCL)'
    elif backend == 'gloo':
        backend = f'gloo ({world} ranks on {torch.cuda.device_count()} device(s): functional check, not a scaling number)'

    if args.device_warmup_seconds > 0:      # device initialisation, not a step: bring clocks and power state up
        a = torch.randn(4096, 4096, device=device)
        t_end = time.perf_counter() + args.device_warmup_seconds
        while time.perf_counter() < t_end:
            for _ in range(20):
                a = (a @ a) * 1e-2
            torch.cuda.synchronize()
        del a

    is_pp = 'pointpillars' in os.path.basename(args.config)
    want_roofline = rank == 0 and not args.no_roofline and is_pp
    sites = []
    if want_roofline:
        from gga_amd import Config
        me = Config.fromfile(args.config).model.pts_middle_encoder
        fh, fw = me.output_shape[0] // 2, me.output_shape[1] // 2
        # one pair per step around the scatter op; one per 128 -> 128 dense 3x3 convolution of SECOND's second stage (half the
        # head's map size; forward and backward-data): the launches of dense_conv3x3_ws_kernel<4, 2> (two planes; three planes:
        # dense_conv3x3_x9_kernel<4, 16, 3, 2, 0>), the instantiation the largest share of the step's time is in
        sites = [(_lib.TIME_SCATTER_FWD, args.steps, 0),
                 (_lib.TIME_DENSE_CONV, 36 * args.steps, _lib.timing_conv_key(128, 128, (fh // 2) * (fw // 2)))]
    main_run = run_workload(args.config, args.batch, args.steps, args.warmup, args, rank, world, device, sites)
    dt = main_run['dt']
    ms_per_step = dt / args.steps * 1e3

    res = None
    if rank == 0:
        res = {
            'metric': 'kitti_frames_per_sec_gga_train_step', 'value': round(args.batch * world * args.steps / dt, 3),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPE_PLANES2, 'arith': ARITH, 'parity': PARITY, 'data': 'synthetic',
            'config': {'workload': ('BASELINE config #2: PointPillars voxelize+PFN+scatter + SECOND/FPN + '
                                    'CenterHead_GGA losses' if is_pp else
                                    'gga_kitti_config.py: voxelize + HardSimpleVFE + SparseEncoder + SECOND/FPN + '
                                    'CenterHead_GGA losses') + ', full train step (fwd+bwd+clip+AdamW)',
                       'config_file': os.path.relpath(args.config, REPO),
                       'frames_per_gpu': args.batch, 'global_batch': args.batch * world, 'points_per_frame': 20000,
                       'parallelism': f'dp{world}', 'backend': backend,
                       'memory_format': 'nchw' if args.nchw else 'channels_last',
                       'final_loss': round(main_run['loss'], 4)},
        }
        tm = main_run['timings']
        if tm.get(_lib.TIME_SCATTER_FWD):
            res['roofline'] = scatter_roofline(main_run['model'], main_run['batches'], tm[_lib.TIME_SCATTER_FWD])
        if tm.get(_lib.TIME_DENSE_CONV):
            flops = 2.0 * args.batch * (fh // 2) * (fw // 2) * 128 * 128 * 9
            kname = 'dense_conv3x3_ws_kernel<4,2>' if main_run['runner'].planes == 2 else 'dense_conv3x3_x9_kernel<4,16,3,2,0>'
            res['mfma_roofline'] = mfma_roofline('%s (128->128, %dx%d, fwd + bwd-data)' % (kname, fh // 2, fw // 2),
                                                 flops, tm[_lib.TIME_DENSE_CONV],
                                                 len(tm[_lib.TIME_DENSE_CONV]) / args.steps, ms_per_step, main_run['runner'].planes)
        res['config']['matrix_planes'] = main_run['runner'].planes          # what the timed steps ran on (2 unless the guard fell back)
        if main_run['runner'].planes != 2:
            res['dtype'] = DTYPE_PLANES3
        res['range_guard'] = main_run['runner'].range_reports
    cfg_main = main_run['cfg']
    parity_pack = None
    if rank == 0 and world == 1 and is_pp and not args.no_cpu_baseline:
        parity_pack = bench_size_gpu_losses(main_run)
    del main_run
    gc.collect()
    torch.cuda.empty_cache()

    if is_pp and not args.no_second_trunk:
        # the reference's shipped model section (sparse-conv trunk), same run, same launch
        s_sites = [(_lib.TIME_SPARSE_CONV, 64 * args.steps, 0), (_lib.TIME_SPARSE_WGRAD, 32 * args.steps, 0)] if rank == 0 else []
        # (Rounds 3-4 warmed this leg up for 20 steps to step over a "first-process transient": its first 15-25 steps ran 53-70 ms
        # on the device. Root cause, round 5 (tools_dev/first_steps.py, malloc_trace.py, who_holds.py): the prefetched index plan
        # hung on the coordinate tensor its first level keeps - a reference cycle; each step's 1.6 GB of levels and rule books
        # waited for a generation-2 pass of Python's collector (every ~12 steps) and the caching allocator answered with ~21
        # hipMalloc calls per step, 48 GB reserved after 30 steps. Without the cycle the pool is complete after 8 steps at
        # 22 GB. One warm-up count for all legs again.)
        sw = max(args.warmup, 8)       # (its allocator pool is complete after 8 steps: with fewer, hipMalloc calls fall into the timed steps - ADVICE r05; reported in `warmup`)
        sec = run_workload(SECOND_CONFIG, args.second_batch, args.steps, sw, args, rank, world, device, s_sites)
        if rank == 0:
            sdt = sec['dt']
            st = sec['timings']
            conv_ms = sum(st.get(_lib.TIME_SPARSE_CONV, [])) / args.steps
            wg_ms = sum(st.get(_lib.TIME_SPARSE_WGRAD, [])) / args.steps
            res['second_trunk'] = {
                'config_file': os.path.relpath(SECOND_CONFIG, REPO),
                'workload': 'voxelize + HardSimpleVFE + SparseEncoder (sparse 3D conv) + SECOND/FPN + CenterHead_GGA '
                            'losses, full train step (fwd+bwd+clip+AdamW)',
                'frames_per_gpu': args.second_batch, 'global_batch': args.second_batch * world,
                'value': round(args.second_batch * world * args.steps / sdt, 3), 'unit': 'frames/s',
                'ms_per_step': round(sdt / args.steps * 1e3, 3), 'steps': args.steps, 'warmup': sw,
                'final_loss': round(sec['loss'], 4),
                'dominant_kernels_ms_per_step': {
                    'sp_conv_x9_kernel + sp_conv_halo_kernel (sparse conv fwd + bwd-data, %d launches/step)' % (len(st.get(_lib.TIME_SPARSE_CONV, [])) // args.steps):
                        round(conv_ms, 3),
                    'sparse conv weight gradient (%d launches/step)' % (len(st.get(_lib.TIME_SPARSE_WGRAD, [])) // args.steps):
                        round(wg_ms, 3)},
                'timed': 'in-step, HIP events on the launch stream'}
        sroof = None if args.no_roofline else sparse_roofline(sec, args)       # on every rank (its steps all-reduce)
        if rank == 0:
            res['second_trunk']['roofline'] = sroof
        del sec
        gc.collect()
        torch.cuda.empty_cache()

    if is_pp and not args.no_pgd:
        lw = args.warmup
        pg = run_mono_workload(args.pgd_batch, args.steps, lw, args, rank, world, device)
        if rank == 0:
            res['pgd_trunk'] = {
                'config_file': os.path.relpath(PGD_CONFIG, REPO),
                'workload': 'FCOSMono3D: ResNet-101 (caffe) + FPN + PGDHead with DCNv2 towers, KITTI-mono3d images 1242x375 '
                            'padded to 1248x384, full train step (fwd+bwd+clip+SGD)',
                'frames_per_gpu': args.pgd_batch, 'global_batch': args.pgd_batch * world,
                'value': round(args.pgd_batch * world * args.steps / pg['dt'], 3), 'unit': 'frames/s',
                'ms_per_step': round(pg['dt'] / args.steps * 1e3, 3), 'steps': args.steps, 'warmup': lw,
                'final_loss': round(pg['loss'], 4),
                'weights': 'random init: the open-mmlab://detectron2/resnet101_caffe checkpoint of the config is not '
                           'available offline (Kaiming backbone with the last norm of every bottleneck at 0.2, head per '
                           'PGDHead.init_weights)'}
        gc.collect()
        torch.cuda.empty_cache()

    if is_pp and not args.no_fcaf3d:
        lw = args.warmup
        fc = run_indoor_workload(args.fcaf3d_batch, args.steps, lw, args, rank, world, device)
        if rank == 0:
            res['fcaf3d_trunk'] = {
                'config_file': os.path.relpath(FCAF3D_CONFIG, REPO),
                'workload': 'MinkSingleStage3DDetector: MinkResNet-34 + FCAF3DHead (generative transposed convolutions, pruning), '
                            'synthetic SUN RGB-D-shaped scenes of 50 000 points, 10 classes, full train step (fwd+bwd+clip+AdamW); '
                            'stock FCAF3D - the reference has no GGA head for this trunk',
                'frames_per_gpu': args.fcaf3d_batch, 'global_batch': args.fcaf3d_batch * world,
                'value': round(args.fcaf3d_batch * world * args.steps / fc['dt'], 3), 'unit': 'scenes/s',
                'ms_per_step': round(fc['dt'] / args.steps * 1e3, 3), 'steps': args.steps, 'warmup': lw,
                'final_loss': round(fc['loss'], 4), 'matrix_planes': fc['runner'].planes}
        del fc
        gc.collect()
        torch.cuda.empty_cache()

    if is_pp and world == 1 and not args.no_loader_fed:
        res['loader_fed'] = run_loader_fed(args, device, res['value'])

    if is_pp and world == 1 and not args.no_inference:
        from gga_amd import synthetic as _syn
        res['inference'] = run_inference(args, device, args.config, _syn.RANGE_PP, frames=args.inference_frames)
        res['inference']['vs_train_step_frames_per_s'] = round(res['inference']['samples_per_gpu_16']['value'] / res['value'], 2)
        if not args.no_second_trunk:
            res['inference']['second_trunk'] = run_inference(args, device, SECOND_CONFIG, _syn.RANGE_SECOND, frames=args.inference_frames)
            res['inference']['second_trunk']['vs_train_step_frames_per_s'] = round(
                res['inference']['second_trunk']['samples_per_gpu_16']['value'] / res['second_trunk']['value'], 2)

    if is_pp and not args.no_planes3:
        # the same steps on the library's default arithmetic (three bf16 planes / six products: fp32 semantics per element)
        s3, w3 = args.steps, args.warmup          # the same protocol as the headline
        p3 = run_workload(args.config, args.batch, s3, w3, args, rank, world, device, planes=3)
        if rank == 0:
            res['planes3'] = {'dtype': DTYPE_PLANES3, 'arith': 'three bf16 planes / six partial products (fp32 exponent range)', 'steps': s3, 'warmup': w3,
                              'ms_per_step': round(p3['dt'] / s3 * 1e3, 3),
                              'value': round(args.batch * world * s3 / p3['dt'], 3), 'unit': 'frames/s',
                              'final_loss': round(p3['loss'], 4)}
        del p3
        gc.collect()
        torch.cuda.empty_cache()
        if not args.no_second_trunk:
            p3 = run_workload(SECOND_CONFIG, args.second_batch, s3, w3, args, rank, world, device, planes=3)
            if rank == 0:
                res['planes3']['second_trunk'] = {'ms_per_step': round(p3['dt'] / s3 * 1e3, 3),
                                                  'value': round(args.second_batch * world * s3 / p3['dt'], 3),
                                                  'unit': 'frames/s', 'final_loss': round(p3['loss'], 4)}
            del p3
            gc.collect()
            torch.cuda.empty_cache()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'], res['parity_at_bench_size'] = cpu_baseline(cfg_main, parity=parity_pack)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
