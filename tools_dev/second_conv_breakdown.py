"""Per-launch breakdown of the gather-GEMM kernels (sparse 3D, strided, transposed convolutions) inside one train step:
rows, taps, channels, real (input row, output row) pairs, microseconds (HIP events around each call, so the launches are
serialised - use for relative shares), and the fp32-equivalent rate on real pairs.
    python tools_dev/second_conv_breakdown.py [second|pp] [batch]"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from gga_amd import _lib, sparse, strided_conv

which = sys.argv[1] if len(sys.argv) > 1 else 'second'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (8 if which == 'second' else 16)
cfg_path = bench.SECOND_CONFIG if which == 'second' else bench.PP_CONFIG
maps = {}


def remember(t):
    if torch.is_tensor(t):
        maps[t.data_ptr()] = t


o1 = sparse._Rulebook.__init__
def init1(self, nbr):
    o1(self, nbr)
    remember(self.nbr)
sparse._Rulebook.__init__ = init1
o2 = strided_conv._Book.__init__
def init2(self, *a, **k):
    o2(self, *a, **k)
    for n in ('fwd', 'bwd'):
        remember(getattr(self, n, None))
strided_conv._Book.__init__ = init2

L = _lib.lib()
records, active = [], [False]
pair_cache = {}


def pairs_of(ptr, n_rows, kvol):
    ptr = getattr(ptr, 'value', ptr)
    t = maps.get(ptr)
    if t is None:
        return None
    key = (ptr, t._version)
    if key not in pair_cache:
        pair_cache[key] = int((t >= 0).sum())
    return pair_cache[key]


def wrap(name, unpack):
    real = getattr(L, name)

    def f(*a):
        if not active[0]:
            return real(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real(*a)
        e1.record()
        records.append((name, unpack(a), e0, e1))
        return r
    setattr(L, name, f)


wrap('gga_sparse_conv_apply_bn_bwd', lambda a: dict(map=a[1], rows=a[5], kvol=a[6], cin=a[7], cout=a[8], flip=a[9], bn=bool(a[17])))
wrap('gga_sparse_conv_apply_stats', lambda a: dict(map=a[1], rows=a[5], kvol=a[6], cin=a[7], cout=a[8], flip=a[9], bn=False))
wrap('gga_sparse_conv_wgrad_planes', lambda a: dict(map=a[4], rows=a[5], kvol=a[6], cin=a[7], cout=a[8], flip=-1, bn=False))

args = bench.parse_args(['--config', cfg_path, '--batch', str(batch), '--steps', '1', '--warmup', '3', '--no-cpu-baseline', '--no-roofline'])
run = bench.run_workload(cfg_path, batch, 1, 3, args, 0, 1, torch.device('cuda:0'))
runner, batches = run['runner'], run['batches']
active[0] = True
runner.step(batches[0])
torch.cuda.synchronize()
active[0] = False
rows = []
for name, d, e0, e1 in records:
    us = e0.elapsed_time(e1) * 1e3
    p = pairs_of(d['map'], d['rows'], d['kvol'])
    kind = 'wgrad' if 'wgrad' in name else ('bwd' if d['flip'] == 1 else 'fwd')
    fl = 2.0 * p * d['cin'] * d['cout'] if p is not None else None
    rows.append(dict(kind=kind, rows=d['rows'], kvol=d['kvol'], cin=d['cin'], cout=d['cout'], pairs=p, us=round(us, 1),
                     tflops_fp32_eq=round(fl / us / 1e6, 1) if fl else None, pairs_per_row=round(p / d['rows'], 2) if p else None))
tot = sum(r['us'] for r in rows)
print(f'{len(rows)} launches, {tot / 1e3:.2f} ms (serialised by the events)')
agg = {}
for r in rows:
    k = (r['kind'], r['rows'], r['kvol'], r['cin'], r['cout'])
    a = agg.setdefault(k, dict(n=0, us=0.0, pairs=r['pairs']))
    a['n'] += 1
    a['us'] += r['us']
print(f'{"kind":6s} {"rows":>8s} {"kvol":>4s} {"cin":>4s} {"cout":>4s} {"pairs/row":>9s} {"n":>3s} {"us each":>8s} {"ms":>7s} {"TF/s fp32-eq":>12s}')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['us']):
    each = a['us'] / a['n']
    tf = 2.0 * a['pairs'] * k[3] * k[4] / each / 1e6 if a['pairs'] else 0
    print(f'{k[0]:6s} {k[1]:8d} {k[2]:4d} {k[3]:4d} {k[4]:4d} {(a["pairs"] or 0) / k[1]:9.2f} {a["n"]:3d} {each:8.1f} {a["us"] / 1e3:7.3f} {tf:12.1f}')
os.makedirs('gpurun_out', exist_ok=True)
json.dump(rows, open(f'gpurun_out/conv_breakdown_{which}.json', 'w'))
