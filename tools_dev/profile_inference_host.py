"""Where the host time of the pseudo-label run goes (samples_per_gpu 16): cProfile of single_gpu_test's loop on the bench tree."""
import cProfile, copy, os, pickle, pstats, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from gga_amd import Config, build_model, synthetic
from gga_amd.apis import single_gpu_test
from gga_amd.cnn import to_channels_last
from gga_amd.loader import build_dataloader, build_dataset
args = bench.parse_args([])
dev = torch.device('cuda:0')
root = bench.bench_tree_root()
info_path, _ = synthetic.write_kitti_tree(root, args.loader_frames, pc_range=synthetic.RANGE_PP)
cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_matching_config.py'))
mcfg = Config.fromfile(bench.PP_CONFIG)
test = dict(cfg.data['test'])
pipe = copy.deepcopy(list(test['pipeline']))
for t in pipe[1]['transforms']:
    if t['type'] == 'PointsRangeFilter':
        t['point_cloud_range'] = list(synthetic.RANGE_PP)
infos = pickle.load(open(info_path, 'rb'))[:544]
sub = os.path.join(root, 'kitti_infos_prof.pkl')
pickle.dump(infos, open(sub, 'wb'))
test.update(data_root=root + '/', ann_file=sub, pipeline=pipe, pcd_limit_range=list(synthetic.RANGE_PP), test_mode=True)
m = mcfg.model
m['train_cfg'] = None
m['pts_middle_encoder']['channels_last'] = True
torch.manual_seed(0)
model = to_channels_last(build_model(m).to(dev)).eval()
ds = build_dataset(test)
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
loader = build_dataloader(ds, samples_per_gpu=16, workers_per_gpu=workers, dist=False, shuffle=False)
marks = {}


def progress(n):
    if n == 32:
        torch.cuda.synchronize(); marks['t0'] = time.perf_counter(); pr.enable()
pr = cProfile.Profile()
res = single_gpu_test(model, loader, dev, progress=progress, planes=2)
torch.cuda.synchronize()
pr.disable()
dt = time.perf_counter() - marks['t0']
print(f'workers {workers}: {(len(res) - 32) / dt:.1f} frames/s, {dt / (len(res) - 32) * 1e3 * 16:.1f} ms per batch of 16')
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
