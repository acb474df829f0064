# FCAF3D leg under the tracer: steady-state kernel table + idle gaps (where the host keeps the device waiting)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
cat > /tmp/fc.py <<P
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
args = bench.parse_args(['--steps', '6', '--warmup', '3', '--no-cpu-baseline'])
r = bench.run_indoor_workload(8, 6, 3, args, 0, 1, torch.device('cuda:0'))
print('ms per step', r['dt'] / 6 * 1e3)
P
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_fc -- python3 /tmp/fc.py > /tmp/fc.log 2>&1; tail -1 /tmp/fc.log
python3 $R/tools_dev/trace_summary.py /tmp/tr_fc --steps 3 --top 60 --out $R/gpurun_out/r04_fcaf3d_bs8_steady_state.csv | cut -c1-150
