#!/usr/bin/env python
"""Per-kernel summary of several rocprofv3 --pmc passes over the same command (one directory per pass):
    python tools_dev/pmc_kernels_summary.py <out.json> <pass_dir> [<pass_dir> ...] -- [kernel-name-substring ...]
Launches are grouped by (kernel name without arguments, grid size); per group: launches per pass, mean duration, and the
mean of every counter found. Derived: cycles = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs), clock = cycles / duration,
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles, hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE counts
half of a wide read stream on gfx950: MI355X_MICROARCH.md, HBM section; both in KB), wave-cycle shares from the SQ counters
(quad-cycles: WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing)."""
import collections, csv, glob, json, re, sys
out = sys.argv[1]
rest = sys.argv[2:]
dirs, subs = (rest[:rest.index('--')], rest[rest.index('--') + 1:]) if '--' in rest else (rest, [])
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            if subs and not any(s in name for s in subs):
                continue
            short = re.sub(r'\(.*', '', name.replace('void ', ''))
            key = (short, int(r['Grid_Size']))
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
            did = (d, r['Dispatch_Id'])
            if did not in seen:
                seen.add(did)
                dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
res = []
for key, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    n = max(len(v) for v in cs.values())
    us = sum(dur[key]) / len(dur[key]) / 1e3
    e = dict(kernel=key[0], grid_threads=key[1], launches_per_pass=n, avg_us=round(us, 1))
    if 'GRBM_GUI_ACTIVE' in m:
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        e.update(cycles=round(cyc), clock_GHz=round(cyc / (us * 1e3), 2))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
            e['mfma_busy'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc, 3)
    if 'FETCH_SIZE' in m or 'WRITE_SIZE' in m:
        hb = (2 * m.get('FETCH_SIZE', 0.0) + m.get('WRITE_SIZE', 0.0)) * 1024
        e.update(hbm_bytes_per_launch=int(hb), hbm_GBps=round(hb / (us * 1e-6) / 1e9, 1))
    if 'SQ_WAVE_CYCLES' in m and m['SQ_WAVE_CYCLES'] > 0:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
            if c in m:
                e[c.lower() + '_share'] = round(m[c] / m['SQ_WAVE_CYCLES'], 3)
    e['counters'] = {c: round(v, 1) for c, v in sorted(m.items())}
    res.append(e)
res.sort(key=lambda r: -r['avg_us'] * r['launches_per_pass'])
json.dump(dict(how=__doc__.split('\n', 3)[3].replace('\n', ' '), passes=dirs, kernels=res), open(out, 'w'), indent=1)
for r in res[:25]:
    print({k: v for k, v in r.items() if k != 'counters'})
