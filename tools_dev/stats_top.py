"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools_dev/stats_top.py <dir> <steps> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot / 1e6 / steps:.1f} ms/step over {steps:.0f} steps')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:n]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.2f} ms/step {int(r['Calls']) / steps:7.1f} calls {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
