"""Which source lines index a gradient-carrying tensor with a tensor (torch's backward for that is the slow sort-based scatter)?
One FCAF3D train step with Tensor.__getitem__ wrapped; prints file:line, count, largest indexed shape."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
args = bench.parse_args(['--steps', '1', '--warmup', '1', '--no-cpu-baseline'])
sites = collections.defaultdict(lambda: [0, None])
orig = torch.Tensor.__getitem__


def spy(self, idx):
    if isinstance(self, torch.Tensor) and self.requires_grad and torch.is_grad_enabled():
        parts = idx if isinstance(idx, tuple) else (idx,)
        if any(isinstance(p, torch.Tensor) and p.dim() > 0 for p in parts):
            st = [f for f in traceback.extract_stack(limit=12) if 'gga_amd' in f.filename or 'bench' in f.filename]
            k = ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in st[-3:][::-1])
            sites[k][0] += 1
            if sites[k][1] is None or self.numel() > sites[k][1][0]:
                sites[k][1] = (self.numel(), tuple(self.shape))
    return orig(self, idx)


torch.Tensor.__getitem__ = spy
bench.run_indoor_workload(8, 1, 1, args, 0, 1, torch.device('cuda:0'))
torch.Tensor.__getitem__ = orig
for k, (n, shp) in sorted(sites.items(), key=lambda kv: -kv[1][1][0]):
    print(k, n // 2, 'per step, largest', shp[1])
