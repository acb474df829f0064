export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
cat > /tmp/pg.py <<P
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
args = bench.parse_args(['--steps', '5', '--warmup', '3', '--no-cpu-baseline'])
r = bench.run_mono_workload(12, 5, 3, args, 0, 1, torch.device('cuda:0'))
print('ms per step', r['dt'] / 5 * 1e3)
P
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_pg -- python3 /tmp/pg.py > /tmp/pg.log 2>&1; tail -1 /tmp/pg.log
python3 $R/tools_dev/trace_summary.py /tmp/tr_pg --steps 3 --top 45 --out $R/gpurun_out/r04_pgd_bs12_steady_state.csv | cut -c1-170
