#!/usr/bin/env python
"""Mean counter values per kernel (name prefix match) from rocprofv3 --pmc counter_collection CSVs."""
import csv, glob, sys, collections
pref = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if pref in r['Kernel_Name']:
                key = r['Kernel_Name'].split('(')[0][-40:]
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})')
