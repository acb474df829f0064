"""Does the shipped config's train_detector run on the synthetic tree learn? total loss every 20 steps (argv[1]: tag)."""
import os, sys, tempfile
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests')]
from gga_amd import Config, build_model, synthetic
from gga_amd.cnn import to_channels_last
from gga_amd.loader import build_dataset
from gga_amd.train import setup_multi_processes, train_detector
case = sys.argv[2] if len(sys.argv) > 2 else 'second'
cfg_name, rng = (('gga_kitti_config.py', synthetic.RANGE_SECOND) if case == 'second' else ('gga_kitti_pointpillars_config.py', synthetic.RANGE_PP))
root = os.path.join(tempfile.gettempdir(), f'gga_dbg_learn_{case}')
info_path, db_path = synthetic.write_kitti_tree(root, int(os.environ.get('FRAMES', '64')), pc_range=rng, db_per_class=100)
cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', cfg_name))
d = cfg.data['train']
d['dataset'].update(data_root=root + '/', ann_file=info_path)
for t in d['dataset']['pipeline']:
    if t['type'] == 'ObjectSample_GGA':
        t['db_sampler'].update(data_root=root + '/', info_path=db_path)
    if 'point_cloud_range' in t:
        t['point_cloud_range'] = list(rng)
cfg.data.update(samples_per_gpu=int(os.environ.get('B', '2')), workers_per_gpu=2, persistent_workers=os.environ.get('PERSIST', '1') == '1')
cfg.runner = dict(type='EpochBasedRunner', max_epochs=int(os.environ.get('EPOCHS', '10')))
cfg.checkpoint_config, cfg.work_dir, cfg.seed = None, None, 0
if os.environ.get('GUARD_EVERY'):
    cfg['gga_range_check_interval'] = int(os.environ['GUARD_EVERY'])
if os.environ.get('THREADS'):
    torch.set_num_threads(int(os.environ['THREADS']))
setup_multi_processes(cfg)
cfg.model.pts_middle_encoder['channels_last'] = True
import random
import numpy as np
SEED = int(os.environ.get('SEED', '11'))
random.seed(SEED), np.random.seed(SEED), torch.manual_seed(SEED)          # set_random_seed of the reference (tools/train.py)
model = build_model(cfg.model)
with torch.no_grad():
    for th in model.pts_bbox_head.task_heads:
        for name in ('reg', 'height', 'dim', 'rot'):
            getattr(th, name)[-1].weight.mul_(0.05)
model = to_channels_last(model.to('cuda:0')).train()
ds = build_dataset(d)
hist = []
first = []


def after(runner, n):
    pass
from gga_amd.train import Runner
real = Runner.step


def step(self, data, next_data=None):
    out = real(self, data, next_data)
    if self.iter <= int(os.environ.get('FIRST', '0')):
        first.append(round(float(out['loss'].detach()), 3))
    if self.iter % 20 == 0:
        lv = {k: float(v) for k, v in out['log_vars'].items()}
        hist.append((self.iter, round(lv['loss'], 2), {k: round(v, 2) for k, v in lv.items() if 'loss' in k and k != 'loss' and v > 5}))
    return out
Runner.step = step
if os.environ.get('NOPREFETCH') == '1':
    Runner.prefetch = lambda self, data: None
train_detector(model, ds, cfg, distributed=False, device=torch.device('cuda:0'))
print(sys.argv[1], case, 'first', first)
print(sys.argv[1], case, [h[:2] for h in hist])
print(sys.argv[1], 'big terms at the end', hist[-1][2], 'at the start', hist[0][2])
