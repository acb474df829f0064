import os, sys, torch, numpy as np
sys.path.insert(0, '.')
from gga_amd import Config, build_model, synthetic
DEV = 'cuda:0'
cfg = Config.fromfile('configs/gga/gga_kitti_config.py')
torch.manual_seed(0)
model = build_model(cfg.model).to(DEV).train()
b = synthetic.make_batch(2, n_points=20000, pc_range=synthetic.RANGE_SECOND)
pts = [p.to(DEV) for p in b['points']]
def st(name, t):
    t = t.detach().float()
    print(f'{name:28s} shape {tuple(t.shape)} finite {bool(torch.isfinite(t).all())} absmax {float(t.abs().max()):.4g} mean {float(t.mean()):.4g}')
v, n, c = model.voxelize(pts)
st('voxels', v); print('num voxels', len(c), 'npts max', int(n.max()), 'min', int(n.min()))
f = model.pts_voxel_encoder(v, n, c); st('vfe', f)
from gga_amd.sparse import SparseConvTensor
enc = model.pts_middle_encoder
x = SparseConvTensor(f, c.int(), enc.sparse_shape, 2)
x = enc.conv_input(x); st('conv_input', x.features)
for i, layer in enumerate(enc.encoder_layers):
    for j, m in enumerate(layer):
        x = m(x) if hasattr(m, 'forward') else x
        st(f'stage{i+1}.{j} n={x.features.shape[0]} {tuple(x.spatial_shape)}', x.features)
x = enc.conv_out(x); st('conv_out', x.features)
d = x.dense(); st('dense', d)
N, C, D, H, W = d.shape
y = model.pts_backbone(d.view(N, C * D, H, W))
for i, t in enumerate(y): st(f'second{i}', t)
z = model.pts_neck(y); st('neck', z[0])
outs = model.pts_bbox_head(z)
for t in range(3):
    for k, vv in outs[t][0].items(): st(f'head{t}.{k}', vv)
data = dict(b, points=pts)
losses = model.pts_bbox_head.loss(data['gt_bboxes_3d'], data['gt_labels_3d'], outs, data['GGA_boxes_img'], data['GGA_lidar2img'], data['GGA_init_pseudo_labels'], data['GGA_bdry_masks'], data['GGA_in_box_points'], data['img_metas'])
print({k: float(v) for k, v in losses.items()})
print('--- neighbour density per level (valid offsets / kvol) ---')
x = SparseConvTensor(f, c.int(), enc.sparse_shape, 2)
x = enc.conv_input(x)
for i, layer in enumerate(enc.encoder_layers):
    for j, m in enumerate(layer):
        x = m(x)
    lv = x._level
    nbr = lv.subm_rulebook((3, 3, 3))
    valid = (nbr >= 0)
    mask = (valid.long() << torch.arange(27, device=nbr.device)[:, None]).sum(0)
    print(f'level after stage {i+1}: n={lv.n} avg valid offsets {float(valid.float().sum(0).mean()):.2f} / 27, distinct masks {len(torch.unique(mask))}')
    srt = torch.sort(mask)[0]
    t = srt[: (len(srt) // 64) * 64].view(-1, 64)
    union = torch.zeros(t.shape[0], dtype=torch.long, device=t.device)
    for b in range(64):
        union |= t[:, b]
    pc = sum(((union >> k) & 1) for k in range(27)).float().mean()
    print(f'   mask-sorted 64-row tiles: avg offsets touched per tile {float(pc):.2f}')
