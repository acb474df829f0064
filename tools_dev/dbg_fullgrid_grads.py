"""tests/test_sparse_gpu.py::test_sparse_encoder_full_grid_vs_pair_list_reference's gradient part, with switches:
   argv[1] = 'bn' | 'nobn' (the BatchNorm-backward epilogue of the backward-data launches on / off), env GGA_SP_OFFSET_SUMS."""
import copy, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests')]
from gga_amd import Config, synthetic, dense_conv, functional as F
from gga_amd.registry import build_middle_encoder
from oracle import sparse_ref as SR, torch_ref as R
mode = sys.argv[1] if len(sys.argv) > 1 else 'bn'
planes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dense_conv.PLANES = planes
if mode == 'nobn':
    dense_conv.bn_source = lambda *a, **k: None
DEV = 'cuda:0'
cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
torch.manual_seed(0)
enc = build_middle_encoder(cfg.model.pts_middle_encoder)
enc.train()
ref, ref64 = copy.deepcopy(enc), copy.deepcopy(enc).double()
B = 4
batch = synthetic.make_batch(B, n_points=20000, pc_range=synthetic.RANGE_SECOND)
vl = cfg.model.pts_voxel_layer
v, n, c, _ = F.hard_voxelize_batch([p.to(DEV) for p in batch['points']], vl.voxel_size, vl.point_cloud_range, vl.max_num_points, vl.max_voxels[0])
feats = F.voxel_mean(v, n, 4)
ref.to(DEV), ref64.to(DEV), enc.to(DEV)
yr, _ = SR.sparse_encoder_reference(ref, feats, c, B, pairs=True)
y = enc(feats, c, B)
g = torch.randn_like(yr)
yr.backward(g)
y.backward(g)
y64, _ = SR.sparse_encoder_reference(ref64, feats.double(), c, B, pairs=True)
y64.backward(g.double())
grads = {n_: p.grad for n_, p in enc.named_parameters()}
rows = R.gradient_offenders(grads, ref, ref64, tol=-1.0, slack=0.0)
print(mode, 'planes', planes, 'offset_sums', os.environ.get('GGA_SP_OFFSET_SUMS', '1'),
      [(n_.replace('encoder_layers.encoder_layer', 'L'), round(e, 5)) for n_, e, f in rows if 'weight' in n_ and ('conv' in n_ or '.0.weight' in n_)])
