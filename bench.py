#!/usr/bin/env python
"""Headline benchmark: GGA train step on synthetic KITTI-shaped frames (BASELINE.json).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full train step of BASELINE config #2 on one batch: batched hard voxelize
(PointPillars grid) -> PillarFeatureNet -> pillar scatter -> SECOND -> SECONDFPN ->
CenterHead_GGA -> GGA losses -> backward -> (DDP all-reduce) -> grad clip -> AdamW, bs=16
frames per GPU, fp32. Point clouds are resident in HBM before the timed region; weak scaling
(every rank steps its own 16 frames, gradients all-reduced over RCCL).

Rank 0 prints ONE JSON line; `roofline` is the pillar scatter (BASELINE's HBM metric) and
`mfma_roofline` the 64 -> 64 dense 3x3 convolution (the step's dominant kernel), both timed with
HIP events inside the timed steps; `cpu_baseline` is the oracle's CPU restatement of the same
step on a bounded sample (N=1 only).
"""
import argparse
import gc
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 achievable


def scatter_roofline(model, batches, device, step_ms):
    """`roofline` of the pillar-scatter canvas kernel: algorithmic bytes over
    its mean duration INSIDE the timed steps (`step_ms`: one HIP-event pair per step around the
    kernel, on the stream it is launched on)."""
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
    # HBM bytes per launch from the PMC passes kept under profiles/ (FETCH_SIZE doubled per the
    # gfx950 note, WRITE_SIZE exact); only quoted when the shape is the profiled one
    traffic = None
    try:
        pmc = json.load(open(os.path.join(REPO, 'profiles', 'r01_scatter_pmc.json')))
        if (B, ch, me.ny, me.nx) == (16, 64, 496, 432) and abs(m - 256000) < 2000:
            traffic = sum(pmc['kernels'][k]['hbm_bytes_corrected'] for k in knames)
    except (OSError, KeyError, ValueError):
        pass
    out = {'bound': 'hbm', 'kernel': '+'.join(knames), 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': traffic, 'algorithmic_bytes': int(algo),
           'kernel_ms': round(ms, 4), 'launches_timed': len(step_ms), 'timed': 'in-step, HIP events on the launch stream',
           'pillars': int(m)}
    return out


def cpu_baseline(cfg, frames=16):
    """The oracle's CPU restatement of the same train step (C voxelizer + torch fp32 on the
    host cores), one timed step on `frames` frames after a 1-frame warm-up."""
    from gga_amd import build_model, synthetic
    from oracle import torch_ref as R
    # torch's CPU convolutions stop scaling (and oversubscribe badly) far below the 256 hardware
    # threads of the GPU host: use at most 32 threads and report that number as `cores`
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = build_model(cfg.model)
    damp_head_init(model, 0.05)
    model.train()
    warm = synthetic.make_batch(1, start=900, n_points=2000, pc_range=synthetic.RANGE_PP, n_obj_range=(2, 3))
    R.reference_train_step(model, warm)      # first-touch / thread-pool warm-up
    model.zero_grad()
    batch = synthetic.make_batch(frames, start=901, pc_range=synthetic.RANGE_PP)
    t0 = time.perf_counter()
    R.reference_train_step(model, batch)
    dt = time.perf_counter() - t0
    return {'value': round(frames / dt, 4), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'1 train step (fwd+bwd, no optimizer) on {frames} synthetic frames, {dt:.1f} s'}


def damp_head_init(model, scale):
    """Random-init only: Kaiming(fan_out) on the 1-3 channel output convs gives log-dimensions of
    std ~5 (boxes of e^10 m and worse) on noise inputs, i.e. inf/NaN losses that say nothing about
    speed. Scale those output convs so the synthetic run stays finite; architecture, shapes and
    every kernel launched are unchanged."""
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(scale)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='frames per GPU')
    ap.add_argument('--config', default=os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
    ap.add_argument('--nchw', action='store_true',
                    help='keep the reference NCHW memory layout for the BEV trunk (default: channels-last memory, '
                         'same logical tensors and values)')
    ap.add_argument('--head-init-scale', type=float, default=0.05,
                    help='damp the random init of the regression heads\' output convs (see damp_head_init)')
    ap.add_argument('--miopen-find', action='store_true', help='torch.backends.cudnn.benchmark=True (MIOpen find mode)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()
    args.channels_last = not args.nchw

    import gga_amd  # noqa: F401
    from gga_amd import Config, build_model, synthetic
    from gga_amd.train import Runner, init_dist

    rank, world, local_rank = init_dist()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (the product has no CPU path)'
    device = torch.device('cuda', local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)

    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    cfg = Config.fromfile(args.config)
    if args.channels_last and cfg.model.pts_middle_encoder.type == 'PointPillarsScatter':
        cfg.model.pts_middle_encoder['channels_last'] = True
    torch.manual_seed(0)
    model = build_model(cfg.model).to(device)
    damp_head_init(model, args.head_init_scale)
    if args.channels_last:
        from gga_amd.cnn import to_channels_last
        model = to_channels_last(model)
    model.train()
    runner = Runner(model, cfg, max_iters=max(1000, args.steps + args.warmup), distributed=world > 1, device=device)

    pc_range = tuple(cfg.model.pts_voxel_layer.point_cloud_range)
    batches = []
    for i in range(2):      # two distinct batches per rank, point clouds resident in HBM
        b = synthetic.make_batch(args.batch, start=i * args.batch, rank=rank, pc_range=pc_range)
        b['points'] = [p.to(device) for p in b['points']]
        batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
    torch.cuda.synchronize()

    for i in range(args.warmup):
        runner.step(batches[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    want_roofline = rank == 0 and not args.no_roofline and hasattr(model.pts_middle_encoder, 'ny')
    if want_roofline:      # one HIP-event pair per step around the scatter canvas kernel (no synchronisation)
        from gga_amd import _lib
        _lib.check(_lib.lib().gga_pillar_scatter_timing_begin(min(args.steps, 256)), 'timing_begin')
        # and around every 64 -> 64 dense 3x3 convolution at the head's map size (forward and backward-data
        # of the first conv of the 15 head branches and of SECOND block 1): the step's dominant kernel
        fh, fw = model.pts_middle_encoder.ny // 2, model.pts_middle_encoder.nx // 2
        _lib.check(_lib.lib().gga_dense_conv3x3_timing_begin(512, 64, 64, fh * fw), 'dense_timing_begin')
    # no generational garbage collection inside the timed region: a gen-2 pass over the module tree
    # takes tens of ms and drains the launch queue (2 of ~12 runs measured +8..16 ms/step before
    # this, 8 of 8 runs 72.5-72.8 ms after)
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = runner.step(batches[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    scatter_ms, dense_ms = [], []
    if want_roofline:
        buf = (C.c_float * 256)()
        n = _lib.lib().gga_pillar_scatter_timing_collect(buf, 256)
        _lib.check(min(n, 0), 'timing_collect')
        scatter_ms = [buf[i] for i in range(n)]
        buf2 = (C.c_float * 512)()
        n2 = _lib.lib().gga_dense_conv3x3_timing_collect(buf2, 512)
        _lib.check(min(n2, 0), 'dense_timing_collect')
        dense_ms = [buf2[i] for i in range(n2)]
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss = float(out['loss'].detach()) if torch.is_tensor(out['loss']) else float(out['loss'])
    assert loss == loss, 'loss is NaN'

    if rank == 0:
        res = {
            'metric': 'kitti_frames_per_sec_gga_train_step', 'value': round(args.batch * world * args.steps / dt, 3),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('BASELINE config #2: PointPillars voxelize+PFN+scatter + SECOND/FPN + '
                                    'CenterHead_GGA losses' if 'pointpillars' in os.path.basename(args.config) else
                                    'gga_kitti_config.py: voxelize + HardSimpleVFE + SparseEncoder + SECOND/FPN + '
                                    'CenterHead_GGA losses') + ', full train step (fwd+bwd+clip+AdamW)',
                       'config_file': os.path.relpath(args.config, REPO),
                       'frames_per_gpu': args.batch, 'global_batch': args.batch * world, 'points_per_frame': 20000,
                       'parallelism': f'dp{world}', 'memory_format': 'channels_last' if args.channels_last else 'nchw',
                       'final_loss': round(loss, 4)},
        }
        if scatter_ms:
            res['roofline'] = scatter_roofline(model, batches, device, scatter_ms)
        if dense_ms:
            # the dominant kernel of the step is matrix-bound: fp32 convolution as six bf16 MFMA products per
            # term, so the bf16 matrix work issued is 6 x the fp32 FLOPs; peak = dense bf16 MFMA (MI355X_MICROARCH.md)
            fh, fw = model.pts_middle_encoder.ny // 2, model.pts_middle_encoder.nx // 2
            flops = 2.0 * args.batch * fh * fw * 64 * 64 * 9
            avg = sum(dense_ms) / len(dense_ms)
            res['mfma_roofline'] = {'bound': 'mfma', 'kernel': 'dense_conv3x3_x9_kernel<2> (64->64, %dx%d, fwd + bwd-data)' % (fh, fw),
                                    'achieved': round(6 * flops / (avg * 1e-3) / 1e12, 1), 'peak': 2500.0, 'unit': 'TFLOP/s',
                                    'frac': round(6 * flops / (avg * 1e-3) / 1e12 / 2500.0, 4),
                                    'fp32_equivalent_tflops': round(flops / (avg * 1e-3) / 1e12, 1),
                                    'kernel_ms': round(avg, 4), 'launches_timed': len(dense_ms),
                                    'timed': 'in-step, HIP events on the launch stream',
                                    'share_of_step': round(sum(dense_ms) / args.steps / (dt / args.steps * 1e3), 3)}
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
