#!/usr/bin/env python
"""Steady-state per-step kernel breakdown from a rocprofv3 --kernel-trace CSV: only the last
`--steps` train steps are counted (the first ones contain MIOpen's algorithm search)."""
import argparse, collections, csv, glob, sys
ap = argparse.ArgumentParser()
ap.add_argument('dir'); ap.add_argument('--steps', type=int, default=3); ap.add_argument('--top', type=int, default=30)
ap.add_argument('--out'); ap.add_argument('--gaps', type=int, default=0, help='list the N longest idle gaps of the window with the kernels around them')
a = ap.parse_args()
import os
f = max(glob.glob(a.dir + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
# one train step ends with the optimizer's kernels (a cluster of multi_tensor_apply launches): the window is
# [end of the optimizer of step N - steps, end of the optimizer of step N]. (Voxelizer launches are no step marker:
# with the point-only front prefetched on a side stream the next step's voxelization runs inside the current step.)
csv.field_size_limit(1 << 30)
# the fused optimizer's kernels where there are any (gradient clipping launches multi_tensor_apply kernels of its own, a second
# cluster per step), any multi_tensor_apply kernel otherwise (SGD)
fused = [r for r in rows if 'FusedOptim' in r['Kernel_Name']]
# no fused optimizer (SGD's foreach kernels look like any other multi-tensor addition, and the camera-only head adds its levels'
# gradients with such launches in the middle of the backward pass): the gradient-norm kernels of the clipping, once per step
if not fused:
    fused = [r for r in rows if 'LpNorm' in r['Kernel_Name']]
opt = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in (fused or [r for r in rows if 'multi_tensor_apply' in r['Kernel_Name']]))
ends = []
for i, (st, en) in enumerate(opt):
    if i + 1 == len(opt) or opt[i + 1][0] - en > 2_000_000:      # > 2 ms to the next optimizer kernel: last one of its step
        ends.append(en)
t1, t0 = ends[-1], ends[-1 - a.steps]
rows = [r for r in rows if int(r['End_Timestamp']) <= t1]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r['Start_Timestamp']) >= t0:
        x = agg[r['Kernel_Name'][:110]]; x[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); x[1] += 1
tot = sum(x[0] for x in agg.values())
# device idle inside the window: the union of the kernels' intervals against the window, and how the idle time splits into gaps
# shorter than 3 us (launch boundaries of a device-bound stream), 3-20 us and longer (the host not keeping up)
iv = sorted((max(int(r['Start_Timestamp']), t0), int(r['End_Timestamp'])) for r in rows if int(r['End_Timestamp']) > t0)
busy, cur_s, cur_e, gaps = 0, None, None, []
for st, en in iv:
    if cur_e is None:
        cur_s, cur_e = st, en
    elif st <= cur_e:
        cur_e = max(cur_e, en)
    else:
        busy += cur_e - cur_s
        gaps.append(st - cur_e)
        cur_s, cur_e = st, en
if cur_e is not None:
    busy += cur_e - cur_s
if a.gaps:
    ivn = sorted((max(int(r['Start_Timestamp']), t0), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in rows if int(r['End_Timestamp']) > t0)
    found, end_so_far, last_name = [], None, None
    for st, en, nm in ivn:
        if end_so_far is not None and st > end_so_far:
            found.append((st - end_so_far, (end_so_far - t0) / 1e6, last_name, nm))
        if end_so_far is None or en > end_so_far:
            end_so_far, last_name = en, nm
    for g, at, before, after in sorted(found, reverse=True)[:a.gaps]:
        print(f'# gap {g/1e3:7.1f} us at {at:7.2f} ms: after {before} | before {after}')
g_small = sum(g for g in gaps if g < 3000); g_mid = sum(g for g in gaps if 3000 <= g < 20000); g_big = sum(g for g in gaps if g >= 20000)
idle_line = (f'# device busy (union of kernel intervals) {busy/1e6/a.steps:.2f} ms/step; idle {((t1-t0)-busy)/1e6/a.steps:.2f} ms/step in {len(gaps)/a.steps:.0f} gaps: '
             f'< 3 us {g_small/1e6/a.steps:.2f} ms ({sum(1 for g in gaps if g < 3000)/a.steps:.0f}), 3-20 us {g_mid/1e6/a.steps:.2f} ms ({sum(1 for g in gaps if 3000 <= g < 20000)/a.steps:.0f}), '
             f'>= 20 us {g_big/1e6/a.steps:.2f} ms ({sum(1 for g in gaps if g >= 20000)/a.steps:.0f})')
lines = [f'# steady state over the last {a.steps} steps: wall {(t1-t0)/1e6/a.steps:.2f} ms/step, kernel time {tot/1e6/a.steps:.2f} ms/step',
         idle_line, 'ms_per_step,percent,calls_per_step,avg_us,kernel']
for k, x in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
    lines.append(f'{x[0]/1e6/a.steps:.3f},{100*x[0]/tot:.2f},{x[1]/a.steps:.1f},{x[0]/x[1]/1e3:.1f},"{k}"')
print('\n'.join(lines))
if a.out:
    open(a.out, 'w').write('\n'.join(lines) + '\n')
