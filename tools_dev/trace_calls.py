"""Per-call durations of the kernels matching a substring in the last step of a rocprofv3 kernel trace:
`python tools_dev/trace_calls.py <trace dir> <substring> [n steps back]` -> duration, grid, workgroup per call."""
import csv, glob, os, sys
d, pat = sys.argv[1], sys.argv[2]
f = max(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
opt = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name']]
# step windows end at the last optimizer launch of a cluster
ends = [opt[i] for i in range(len(opt)) if i + 1 == len(opt) or opt[i + 1] - opt[i] > 50]
lo, hi = ends[-2] + 1, ends[-1]
for r in rows[lo:hi]:
    if pat in r['Kernel_Name']:
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {r['Grid_Size_X']:>9}x{r['Grid_Size_Y']}  wg {r['Workgroup_Size_X']}  {r['Kernel_Name'][:60]}")
