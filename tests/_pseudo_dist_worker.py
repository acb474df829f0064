"""Worker of tests/test_loader.py::test_pseudo_label_generation_over_two_ranks: one rank of a 2-rank gloo group on ONE GPU running
``apis.generate_pseudo_labels(distributed=True)`` on the synthetic tree (argv: tree root, checkpoint, raw-output pickle, pseudo-label
file); rank 0 writes both files."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import pickle

import torch
import torch.distributed as dist

from gga_amd.apis import generate_pseudo_labels
from gga_amd.train import init_dist
from test_loader import GOLDEN, matching_cfg

root, ck, raw, out_file, collect = sys.argv[1:6]
rank, world, _ = init_dist()
torch.cuda.set_device(0)
infos = pickle.load(open(os.path.join(GOLDEN, 'gt_database_infos.pkl'), 'rb'))
cfg = matching_cfg(root, infos, os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
cfg.data['test_dataloader'] = dict(samples_per_gpu=1, workers_per_gpu=0)
outputs, res = generate_pseudo_labels(cfg, ck, out=raw, eval_metrics=('mAP',), eval_options=dict(pseudo_label_file=out_file),
                                      device='cuda:0', distributed=True, gpu_collect=collect == 'gpu')
assert (outputs is None) == (rank != 0)
if rank == 0:
    print('PSEUDO frames', len(outputs), res, flush=True)
dist.barrier()
dist.destroy_process_group()
