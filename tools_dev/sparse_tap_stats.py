"""How much of the tap work of the sparse convolutions is real: per rule book of the SECOND trunk (bench frames, bs 8) the
rows, the mean number of present neighbours per row, and the taps a 32-row wave / 128-row workgroup tile executes in the
mask-sorted order (union of its rows' masks)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from gga_amd import sparse

seen = []
orig = sparse._Rulebook.__init__
def init(self, nbr):
    orig(self, nbr)
    seen.append(self)
sparse._Rulebook.__init__ = init
args = bench.parse_args(['--config', bench.SECOND_CONFIG, '--batch', '8', '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-roofline'])
r = bench.run_workload(bench.SECOND_CONFIG, 8, 1, 1, args, 0, 1, torch.device('cuda:0')) if hasattr(bench, 'run_workload') else None
done = set()
for rb in seen[-12:]:
    kvol, n = rb.nbr.shape
    if rb.mask is None or (kvol, n) in done:
        continue
    done.add((kvol, n))
    m = rb.mask[rb.perm.long()].to(torch.int64) & 0xFFFFFFFF
    pop = lambda v: sum(((v >> b) & 1) for b in range(kvol)).float()
    per_row = float(pop(m).mean())
    def union(g):
        k = (n + g - 1) // g * g
        mm = torch.cat([m, m.new_zeros(k - n)]).view(-1, g)
        u = mm[:, 0].clone()
        for j in range(1, g):
            u |= mm[:, j]
        return float(pop(u).mean())
    print(f'kvol {kvol:3d} rows {n:8d}  present taps/row {per_row:5.2f}  executed per 32-row wave {union(32):5.2f}  per 128-row tile {union(128):5.2f}', flush=True)
    # alternative processing orders: by number of taps first; by the bit-reversed mask
    raw = rb.mask.to(torch.int64) & 0xFFFFFFFF
    pc = sum(((raw >> b) & 1) for b in range(kvol))
    for name, key in (('popcount, mask', pc * (1 << 32) + raw),):
        m = raw[torch.sort(key, stable=True)[1]]
        print(f'      order by {name}: per 32-row wave {union(32):5.2f}  per 128-row tile {union(128):5.2f}', flush=True)
