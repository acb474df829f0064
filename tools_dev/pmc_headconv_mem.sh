cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/pmc
i=0
for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc/m$i -- python3 $R/tools_dev/exp_headconv.py > /tmp/pmc/mlog$i.txt 2>&1 || echo "pass $i failed: $c"
done
python3 - <<'P'
import csv, glob, collections
for d in ('m1','m2','m3'):
    for f in glob.glob('/tmp/pmc/%s/**/*counter_collection.csv' % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if 'headconv' not in k: continue
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in agg:
            print(d, k[:60], {c: round(v / cnt[(k, c)], 1) for c, v in agg[k].items()})
P
