#!/usr/bin/env python
"""``python tools/generate_pseudo_labels_gga.py <config> <checkpoint> [--out results.pkl] [--eval mAP] [--eval-options k=v ...]``
- the command of the reference's recipe (README.md:187-192, tools/generate_pseudo_labels_gga.py) for one MI355X: the trained
detector runs over ``cfg.data.test`` and the matching dataset writes the pseudo-label file."""
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
    args = ap.parse_args()
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
    _, result = generate_pseudo_labels(cfg, args.checkpoint, out=args.out, eval_metrics=args.eval, eval_options=opts,
                                       device=f'cuda:{args.gpu_id}', progress=lambda n: print(f'\r{n} frames', end='', flush=True))
    print()
    if result is not None:
        print(result)


if __name__ == '__main__':
    main()
