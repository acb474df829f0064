"""PMC target: a few train steps of configs/gga/gga_kitti_config.py (the sparse trunk, bs 8) and nothing else.
    rocprofv3 --pmc <counters> -d <dir> -- python3 tools_dev/pmc_target_second.py [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
args = bench.parse_args(['--batch', '8', '--steps', str(steps), '--warmup', '2', '--no-cpu-baseline', '--no-roofline'])
bench.run_workload(bench.SECOND_CONFIG, 8, steps, 2, args, 0, 1, torch.device('cuda:0'))
torch.cuda.synchronize()
