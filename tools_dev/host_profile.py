"""Host-side cost of one train step: cProfile over a few steps of the bench workload (the GPU runs
asynchronously; what is measured is the Python / launch time that must stay below the GPU time)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import gga_amd  # noqa: F401
from gga_amd import Config, build_model, synthetic
from gga_amd.cnn import to_channels_last
from gga_amd.train import Runner
import bench

dev = torch.device('cuda:0')
SECOND = len(sys.argv) > 1 and sys.argv[1] == 'second'
cfg = Config.fromfile(bench.SECOND_CONFIG if SECOND else bench.PP_CONFIG)
if not SECOND:
    cfg.model.pts_middle_encoder['channels_last'] = True
BS = 8 if SECOND else 16
torch.manual_seed(0)
model = to_channels_last(build_model(cfg.model).to(dev))
bench.damp_head_init(model, 0.01)
model.train()
runner = Runner(model, cfg, max_iters=1000, distributed=False, device=dev)
pc_range = tuple(cfg.model.pts_voxel_layer.point_cloud_range)
batches = []
for i in range(2):
    b = synthetic.make_batch(BS, start=i * BS, rank=0, pc_range=pc_range)
    b['points'] = [p.to(dev) for p in b['points']]
    batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
for i in range(3):
    runner.step(batches[i % 2])
torch.cuda.synchronize()
# host time per step with the GPU idle at the start of every step (no queue back-pressure)
ts = []
for i in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print('host ms per step (launch side only):', [round(t * 1e3, 1) for t in ts])
pr = cProfile.Profile()
pr.enable()
for i in range(4):
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
if os.environ.get('HOST_PROFILE_TOTTIME'):
    pstats.Stats(pr).sort_stats('tottime').print_stats(40)
