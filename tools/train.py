#!/usr/bin/env python
"""``python tools/train.py <config> [--work-dir D] [--resume-from F] [--auto-resume] [--seed N] [--launcher none|pytorch]`` -
the train entry of the reference (tools/train.py:118-263, started per GPU by tools/dist_train.sh:10-20 through
``torch.distributed.launch``) for one process per MI355X:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py configs/gga/gga_kitti_config.py --launcher pytorch

Config keys read: ``model``, ``data.train`` (+ ``samples_per_gpu`` / ``workers_per_gpu``), ``optimizer``, ``optimizer_config``,
``lr_config``, ``momentum_config``, ``runner.max_epochs``, ``checkpoint_config``, ``log_config.interval``, ``work_dir``,
``resume_from`` / ``load_from``, ``seed``. Validation hooks are not part of the path (``--no-validate`` is the only mode)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description='Train a detector')
    ap.add_argument('config')
    ap.add_argument('--work-dir')
    ap.add_argument('--resume-from')
    ap.add_argument('--auto-resume', action='store_true')
    ap.add_argument('--no-validate', action='store_true', help='(always on: evaluation hooks are out of scope)')
    ap.add_argument('--gpu-id', type=int, default=0)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--diff-seed', action='store_true', help='a different seed per rank')
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    ap.add_argument('--local_rank', '--local-rank', type=int, default=0)
    ap.add_argument('--autoscale-lr', action='store_true')
    args = ap.parse_args()
    os.environ.setdefault('LOCAL_RANK', str(args.local_rank))

    import numpy as np
    import torch
    from gga_amd import Config, build_model
    from gga_amd.cnn import to_channels_last
    from gga_amd.loader import build_dataset
    from gga_amd.train import init_dist, setup_multi_processes, train_detector

    cfg = Config.fromfile(args.config)
    setup_multi_processes(cfg)
    cfg.work_dir = args.work_dir or cfg.get('work_dir') or os.path.join('./work_dirs', os.path.splitext(os.path.basename(args.config))[0])
    if args.resume_from:
        cfg.resume_from = args.resume_from
    if args.auto_resume:
        cfg.auto_resume = True
    distributed = args.launcher != 'none'
    rank, world, local_rank = init_dist() if distributed else (0, 1, args.gpu_id)
    cfg.gpu_ids = list(range(world)) if distributed else [args.gpu_id]
    if args.autoscale_lr:
        cfg.optimizer['lr'] = cfg.optimizer['lr'] * len(cfg.gpu_ids) / 8
    device = torch.device('cuda', local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)
    os.makedirs(cfg.work_dir, exist_ok=True)
    log_path = os.path.join(cfg.work_dir, time.strftime('%Y%m%d_%H%M%S') + '.log')

    def logger(msg):
        if rank == 0:
            print(msg, flush=True)
            with open(log_path, 'a') as f:
                f.write(msg + '\n')

    # --diff-seed offsets the seed of this rank's generators (python, numpy, torch: weight init is broadcast by DDP, the
    # augmentations differ per rank). The SAMPLER's seed must be the same on every rank - each rank takes its block of ONE
    # shuffled order (mmdet syncs it from rank 0, `sync_random_seed`); with the launcher's common --seed that is args.seed.
    seed = args.seed + (rank if args.diff_seed else 0)          # set_random_seed of the reference: python, numpy, torch
    from gga_amd.train import set_random_seed
    set_random_seed(seed, deterministic=getattr(args, 'deterministic', False))
    cfg.seed = args.seed
    cfg.worker_seed = seed
    logger(f'Distributed training: {distributed} ({world} rank(s)); seed {seed}; work_dir {cfg.work_dir}')

    mcfg = cfg.model
    if mcfg.get('pts_middle_encoder', {}).get('type') in ('PointPillarsScatter', 'SparseEncoder'):
        mcfg['pts_middle_encoder']['channels_last'] = True        # the layout the matrix kernels take (same values)
    model = build_model(mcfg, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    if hasattr(model, 'init_weights'):
        model.init_weights()
    dataset = build_dataset(cfg.data['train'])
    model.CLASSES = dataset.CLASSES
    model = to_channels_last(model.to(device))
    train_detector(model, dataset, cfg, distributed=distributed, validate=False, logger=logger, device=device)


if __name__ == '__main__':
    main()
