# L2 / HBM traffic of dense_conv3x3_ws_kernel stand-alone (tools_dev/pmc_target_ws.py): FETCH_SIZE, WRITE_SIZE, L2 requests / hits / misses.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmcwt; mkdir -p /tmp/pmcwt
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcwt/s$i -- python3 $R/tools_dev/pmc_target_ws.py > /tmp/pmcwt/log$i.txt 2>&1 || echo "pass $i failed: $c"
done
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
csv.field_size_limit(1 << 30)
for f in glob.glob('/tmp/pmcwt/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dense_conv3x3_ws' in r['Kernel_Name']:
            key = r['Kernel_Name'][:40] + ' grid ' + r['Grid_Size']
            rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in rows.items():
    print(k, {c: round(sum(v[2:]) / max(len(v[2:]), 1), 1) for c, v in d.items()})
PY
