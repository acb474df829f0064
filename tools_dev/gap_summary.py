#!/usr/bin/env python
"""Idle gaps between consecutive kernels in a rocprofv3 --kernel-trace CSV (last `--steps` train steps),
attributed to the (previous kernel -> next kernel) pair: where the GPU waits for the host."""
import argparse, collections, csv, glob
ap = argparse.ArgumentParser(); ap.add_argument('dir'); ap.add_argument('--steps', type=int, default=3); ap.add_argument('--top', type=int, default=20)
a = ap.parse_args()
rows = list(csv.DictReader(open(glob.glob(a.dir + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [int(r['Start_Timestamp']) for r in rows if r['Kernel_Name'].startswith('vox_insert_kernel')]
opt = [int(r['End_Timestamp']) for r in rows if 'multi_tensor_apply' in r['Kernel_Name'] or 'FusedOptimizer' in r['Kernel_Name']]
t1 = max(opt)
starts = [t for t in starts if t < t1]
t0 = starts[-a.steps]
sel = [r for r in rows if t0 <= int(r['Start_Timestamp']) <= t1]
gaps = collections.defaultdict(lambda: [0, 0]); prev_end = prev = None; tg = tk = 0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev_end is not None and s > prev_end:
        k = (prev[:48], r['Kernel_Name'][:48]); gaps[k][0] += s - prev_end; gaps[k][1] += 1; tg += s - prev_end
    tk += e - s
    if prev_end is None or e > prev_end: prev_end, prev = e, r['Kernel_Name']
print(f'wall {(t1 - t0) / 1e6 / a.steps:.2f} ms/step, kernels {tk / 1e6 / a.steps:.2f} ms/step, idle {tg / 1e6 / a.steps:.2f} ms/step, {len(sel) / a.steps:.0f} kernels/step')
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print(f'{v[0] / 1e6 / a.steps:7.3f} ms/step  n={v[1] / a.steps:5.1f}  avg {v[0] / v[1] / 1e3:8.1f} us   {k[0]}  ->  {k[1]}')
