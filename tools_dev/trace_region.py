"""List every kernel (and memory copy) of a rocprofv3 trace in a time region of the last step: offset from the previous optimizer end,
duration, queue / stream, name. Usage: trace_region.py <dir> <from_ms> <to_ms>"""
import csv, glob, os, sys
d, a, b = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
csv.field_size_limit(1 << 30)
kf = max(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(kf)))
opt = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'FusedOptim' in r['Kernel_Name'])
ends = [en for i, (st, en) in enumerate(opt) if i + 1 == len(opt) or opt[i + 1][0] - en > 2_000_000]
t0 = ends[-2]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'q' + r.get('Queue_Id', '?') + ' s' + r.get('Stream_Id', '?'), r['Kernel_Name'][:70]) for r in rows]
mf = glob.glob(d + '/**/*_memory_copy_trace.csv', recursive=True)
if mf:
    for r in csv.DictReader(open(max(mf, key=os.path.getmtime))):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy s' + r.get('Stream_Id', '?'), r['Direction']))
prev_end = None
for st, en, q, nm in sorted(ev):
    off = (st - t0) / 1e6
    if a <= off <= b:
        gap = '' if prev_end is None or st <= prev_end else f'   <-- idle {(st - prev_end) / 1e3:.0f} us'
        print(f'{off:8.3f} ms {(en - st) / 1e3:7.1f} us  {q:12s} {nm}{gap}')
    prev_end = en if prev_end is None else max(prev_end, en)
