"""A few train steps of configs/gga/gga_pdg.py on synthetic batches (profiling target)."""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
args = bench.parse_args(['--steps', sys.argv[1] if len(sys.argv) > 1 else '3', '--warmup', '2'])
r = bench.run_mono_workload(12, args.steps, args.warmup, args, 0, 1, torch.device('cuda:0'))
print('ms/step', r['dt'] / args.steps * 1e3, 'loss', r['loss'])
