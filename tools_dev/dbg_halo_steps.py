"""Per-step total loss of the shipped config's bench batch with the halo form off / on (same seeds): the trajectories must start
together (the two kernels differ by fp32 summation order) - how fast they part tells how chaotic the synthetic run is."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from gga_amd import sparse
res = {}
for halo in tuple(int(v) for v in os.environ.get("GGA_DBG_SEQ", "0,1,0").split(",")):
    sparse.HALO = halo
    args = bench.parse_args(['--batch', '8', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-roofline'])
    torch.manual_seed(0)
    run = bench.run_workload(bench.SECOND_CONFIG, 8, 1, 0, args, 0, 1, torch.device('cuda:0'))
    losses = [run['loss']]
    for i in range(1, 14):
        out = run['runner'].step(run['batches'][i % 2], next_data=run['batches'][(i + 1) % 2])
        losses.append(float(out['loss']))
    print('halo', halo, ' '.join(f'{l:.6g}' for l in losses))
    del run
    torch.cuda.empty_cache()
