"""For a rocprofv3 --kernel-trace --memory-copy-trace directory: the idle gaps (no kernel running) of the last steps and the
memory copies that overlap each of them. Usage: gap_copies.py <dir> [min_gap_us]"""
import csv, glob, os, sys
d = sys.argv[1]; ming = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
csv.field_size_limit(1 << 30)
kf = max(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
mf = glob.glob(d + '/**/*_memory_copy_trace.csv', recursive=True)
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in csv.DictReader(open(kf)))
cs = []
if mf:
    rows = list(csv.DictReader(open(max(mf, key=os.path.getmtime))))
    print('copy columns:', list(rows[0].keys()) if rows else None)
    for r in rows:
        cs.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', r.get('Name', '?')), r.get('Size', r.get('Bytes', '?'))))
t_end = ks[-1][1]
t0 = t_end - 100_000_000            # the last 100 ms
end_so_far, last = None, None
for st, en, nm in ks:
    if en < t0:
        end_so_far, last = max(end_so_far or en, en), nm
        continue
    if end_so_far is not None and st - end_so_far > ming * 1000:
        ov = [(c[2], c[3], round((c[1] - c[0]) / 1e3, 1)) for c in cs if c[0] < st and c[1] > end_so_far]
        print(f'gap {(st - end_so_far) / 1e3:7.1f} us at {(end_so_far - t0) / 1e6:7.2f} ms: after {last} | before {nm} | copies overlapping: {ov}')
    if end_so_far is None or en > end_so_far:
        end_so_far, last = en, nm
