#!/usr/bin/env python
"""``python tools/generate_pseudo_labels_gga.py <config> <checkpoint> [--out results.pkl] [--eval mAP] [--eval-options k=v ...]``
- the command of the reference's recipe (README.md:187-192, tools/generate_pseudo_labels_gga.py): the trained detector runs over
``cfg.data.test`` and the matching dataset writes the pseudo-label file. One MI355X, or - as tools/dist_pseudo.sh:11-22 of the
reference starts it - one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/generate_pseudo_labels_gga.py <config> <checkpoint> --eval mAP --launcher pytorch"""
import argparse
import ast
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config')
    ap.add_argument('checkpoint')
    ap.add_argument('--out')
    ap.add_argument('--eval', nargs='+')
    ap.add_argument('--eval-options', nargs='+', default=[], help='key=value arguments of dataset.evaluate()')
    ap.add_argument('--gpu-id', type=int, default=0)
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    ap.add_argument('--tmpdir', help='shared directory the ranks\' results are collected through (default: a temporary one)')
    ap.add_argument('--gpu-collect', action='store_true', help='collect the ranks\' results through the process group instead')
    ap.add_argument('--local_rank', '--local-rank', type=int, default=0)
    args = ap.parse_args()
    os.environ.setdefault('LOCAL_RANK', str(args.local_rank))
    assert args.out or args.eval, 'nothing to do: give --out and / or --eval'
    from gga_amd import Config
    from gga_amd.apis import generate_pseudo_labels
    from gga_amd.train import setup_multi_processes
    cfg = Config.fromfile(args.config)
    setup_multi_processes(cfg)
    opts = {}
    for kv in args.eval_options:
        k, v = kv.split('=', 1)
        try:
            opts[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            opts[k] = v
    distributed = args.launcher != 'none'
    rank, gpu = 0, args.gpu_id
    if distributed:
        import torch
        from gga_amd.train import init_dist
        rank, _, local_rank = init_dist()
        gpu = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(gpu)
    progress = (lambda n: print(f'\r{n} frames', end='', flush=True)) if rank == 0 else None
    _, result = generate_pseudo_labels(cfg, args.checkpoint, out=args.out, eval_metrics=args.eval, eval_options=opts,
                                       device=f'cuda:{gpu}', progress=progress, distributed=distributed, tmpdir=args.tmpdir,
                                       gpu_collect=args.gpu_collect)
    if rank == 0:
        print()
        if result is not None:
            print(result)
    if distributed:
        # rank 0 has evaluated (the long part) by now: the other ranks wait here instead of leaving the process group while it is
        # still in use, and the group is torn down in order
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
