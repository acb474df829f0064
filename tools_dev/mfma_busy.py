#!/usr/bin/env python
"""Matrix-pipe utilisation per kernel from a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES` run:

    python tools_dev/mfma_busy.py <pmc_dir> <out.json> [kernel-name-substring ...]

GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles = value / 8); SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
(mfma_busy = value / 1024 / cycles); clock = cycles / duration (MI355X_MICROARCH.md, DVFS note). Launches are grouped
by (kernel, grid size), the first launch of every group (cold caches, clock ramp) is dropped."""
import collections, csv, glob, json, re, sys
d, out = sys.argv[1:3]
subs = sys.argv[3:] or ['dense_conv3x3_x9_kernel', 'dense_wgrad3x3_x9_kernel', 'sp_conv_x9_kernel', 'sp_conv_wgrad_x9_kernel']
csv.field_size_limit(1 << 30)
rows = collections.defaultdict(dict)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if not any(s in name for s in subs):
            continue
        short = re.sub(r'\(.*', '', name.replace('void ', ''))
        key = (short, int(r['Grid_Size']), int(r['Dispatch_Id']))
        rows[key][r['Counter_Name']] = float(r['Counter_Value'])
        rows[key]['_dur'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
groups = collections.defaultdict(list)
for (short, grid, disp), v in sorted(rows.items(), key=lambda kv: kv[0][2]):
    if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        groups[(short, grid)].append(v)
res = []
for (short, grid), vs in groups.items():
    vs = vs[1:] if len(vs) > 1 else vs
    cyc = sum(v['GRBM_GUI_ACTIVE'] for v in vs) / len(vs) / 8
    busy = sum(v['SQ_VALU_MFMA_BUSY_CYCLES'] for v in vs) / len(vs) / 1024
    dur = sum(v['_dur'] for v in vs) / len(vs)
    res.append(dict(kernel=short, grid_threads=grid, launches=len(vs), avg_us=round(dur / 1e3, 1), cycles=round(cyc),
                    clock_GHz=round(cyc / dur, 2), mfma_busy=round(busy / cyc, 3)))
res.sort(key=lambda r: -r['avg_us'] * r['launches'])
json.dump(dict(how=__doc__.split('\n\n')[2].replace('\n', ' '), source=d, kernels=res), open(out, 'w'), indent=1)
for r in res:
    print(r)
