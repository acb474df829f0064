"""Host-side cost of one PGD train step (configs/gga/gga_pdg.py, bs 12): launch-side time per step with the GPU idle at the
start of every step, then a cProfile over a few steps."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench

args = bench.parse_args(['--steps', '2', '--warmup', '2'])
r = bench.run_mono_workload(12, args.steps, args.warmup, args, 0, 1, torch.device('cuda:0'))
runner, batches = r['runner'], r['batches']
torch.cuda.synchronize()
ts = []
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.step(batches[i % len(batches)])
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print('host ms per step (launch side only):', [round(t * 1e3, 1) for t in ts])
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    runner.step(batches[i % len(batches)])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(60)
