"""Uninitialised-read detector: every torch.empty / empty_like / new_empty of the run returns memory filled with NaN (floats) or a
large garbage pattern (integers, bytes). A kernel that reads what it was supposed to write first turns the losses into NaN or
changes them; compare with the plain run (same seeds)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
POISON = os.environ.get('GGA_POISON', '1') == '1'
_e, _el = torch.empty, torch.empty_like


def _fill(t):
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float('nan'))
        elif t.dtype == torch.uint8:
            t.fill_(0xA5)
        elif t.dtype in (torch.int32, torch.int64, torch.int16):
            t.fill_(0x5A5A5A5A if t.dtype != torch.int16 else 0x5A5A)
    return t


if POISON:
    torch.empty = lambda *a, **k: _fill(_e(*a, **k))
    torch.empty_like = lambda *a, **k: _fill(_el(*a, **k))
    _ne = torch.Tensor.new_empty
    torch.Tensor.new_empty = lambda self, *a, **k: _fill(_ne(self, *a, **k))
cfgs = {'second': bench.SECOND_CONFIG, 'pp': bench.PP_CONFIG if hasattr(bench, 'PP_CONFIG') else None}
which = os.environ.get('GGA_POISON_CFG', 'second')
args = bench.parse_args(['--batch', '8', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-roofline'])
torch.manual_seed(0)
cfg = cfgs[which] or args.config
run = bench.run_workload(cfg, 8 if which == 'second' else 16, 1, 0, args, 0, 1, torch.device('cuda:0'))
losses = [run['loss']]
for i in range(1, 6):
    out = run['runner'].step(run['batches'][i % 2], next_data=run['batches'][(i + 1) % 2])
    losses.append(float(out['loss']))
print(which, 'poison' if POISON else 'plain ', ' '.join(f'{l:.9g}' for l in losses))
