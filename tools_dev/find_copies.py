import os, sys, collections, traceback
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools_dev')
import torch
from torch.profiler import profile, ProfilerActivity
import bench
args = bench.parse_args(['--steps', '1', '--warmup', '2'])
# run warmup then profile one step with stacks
import types
orig = bench.run_mono_workload
r = bench.run_mono_workload(12, 1, 2, args, 0, 1, torch.device('cuda:0'))
runner, batches = r.get('runner'), r.get('batches')
print('keys', list(r.keys()))
if runner is None: sys.exit(0)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    runner.step(batches[0])
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.name in ('aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::_to_copy')]
cnt = collections.Counter()
tim = collections.Counter()
for e in ev:
    if e.name != 'aten::copy_': continue
    st = [s for s in (e.stack or []) if 'gga_amd' in s or 'bench' in s]
    key = (st[0] if st else 'autograd/backward (no python frame)', str(e.input_shapes)[:60])
    cnt[key] += 1; tim[key] += e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total
for k, v in sorted(tim.items(), key=lambda kv: -kv[1])[:25]:
    print(f'{v/1e3:8.2f} ms  n={cnt[k]:4d}  {k[0][-90:]}  {k[1]}')
