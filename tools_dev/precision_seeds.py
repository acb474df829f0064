"""How far is each arithmetic from the float64 step, over several seeds? Forward + losses of both LiDAR configs on:
the fp32 CPU restatement, the GPU on two planes, on three planes, and mixed forms (a subsystem on three planes, the rest
on two) - worst relative deviation of the 18 loss entries from the float64 CPU restatement (same weights, batch, SRL
draws). Usage: precision_seeds.py [second|pp] seed [seed ...]"""
import copy, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, synthetic, dense_conv
from gga_amd.cnn import to_channels_last
from oracle import torch_ref as R
DEV = 'cuda:0'
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
case = sys.argv[1]
seeds = [int(s) for s in sys.argv[2:]] or [3]
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'gga', 'gga_kitti_config.py' if case == 'second' else 'gga_kitti_pointpillars_config.py'))
torch.set_num_threads(min(os.cpu_count(), 32))
rel = lambda a, b: abs(a - b) / max(abs(b), 1.0)


def cpu_run(model, batch, srl, dtype):
    m = copy.deepcopy(model).to(dtype)
    losses, _ = R.reference_train_step(m, batch, srl=srl, backward=False)
    return {k: float(v) for k, v in losses.items()}


def gpu_run(model, batch, srl, default, three=()):
    """``three``: attribute paths of the sub-modules that run on three planes while the rest runs on ``default``."""
    dense_conv.PLANES = default
    m = copy.deepcopy(model)
    m.pts_middle_encoder.channels_last = True
    m = to_channels_last(m.to(DEV))
    hs = []
    for path in three:
        mod = m
        for a in path.split('.'):
            mod = getattr(mod, a)
        hs.append(mod.register_forward_pre_hook(lambda mod, inp: setattr(dense_conv, 'PLANES', 3)))
        hs.append(mod.register_forward_hook(lambda mod, inp, out: setattr(dense_conv, 'PLANES', default)))
    data = dict(batch, points=[p.to(DEV) for p in batch['points']])
    with torch.no_grad():
        feats = m.extract_feat(data['points'], None, data['img_metas'])[1]
        outs = m.pts_bbox_head(feats)
        losses = m.pts_bbox_head.loss(data['gt_bboxes_3d'], data['gt_labels_3d'], outs, data['GGA_boxes_img'], data['GGA_lidar2img'],
                                      data['GGA_init_pseudo_labels'], data['GGA_bdry_masks'], data['GGA_in_box_points'], data['img_metas'], srl=srl)
    dense_conv.PLANES = default
    return {k: float(v) for k, v in losses.items()}


rows = []
for seed in seeds:
    torch.manual_seed(seed)
    model = build_model(cfg.model)
    model.train()
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    B = 2
    kw = (dict(n_points=20000, pc_range=synthetic.RANGE_SECOND) if case == 'second' else dict(n_points=5000, pc_range=synthetic.RANGE_PP))
    batch = synthetic.make_batch(B, start=50 + 7 * (seed - 3 if case == 'second' else seed - 1), n_obj_range=(4, 8), n_ibp_range=(10, 200), **kw)
    srl = model.pts_bbox_head.draw_srl(B)
    l64 = cpu_run(model, batch, srl, torch.float64)
    runs = {'cpu32': cpu_run(model, batch, srl, torch.float32),
            'gpu2': gpu_run(model, batch, srl, 2), 'gpu3': gpu_run(model, batch, srl, 3)}
    if case == 'second':
        runs['sparse3'] = gpu_run(model, batch, srl, 2, ('pts_middle_encoder',))
        runs['trunk3'] = gpu_run(model, batch, srl, 2, ('pts_backbone', 'pts_neck'))
        runs['head3'] = gpu_run(model, batch, srl, 2, ('pts_bbox_head',))
        runs['sparse2'] = gpu_run(model, batch, srl, 3, ())  # placeholder replaced below
        # the sparse encoder on two planes, everything else on three
        dense_conv.PLANES = 3
        m = copy.deepcopy(model)
        m.pts_middle_encoder.channels_last = True
        m = to_channels_last(m.to(DEV))
        h1 = m.pts_middle_encoder.register_forward_pre_hook(lambda mod, inp: setattr(dense_conv, 'PLANES', 2))
        h2 = m.pts_middle_encoder.register_forward_hook(lambda mod, inp, out: setattr(dense_conv, 'PLANES', 3))
        data = dict(batch, points=[p.to(DEV) for p in batch['points']])
        with torch.no_grad():
            feats = m.extract_feat(data['points'], None, data['img_metas'])[1]
            outs = m.pts_bbox_head(feats)
            ls = m.pts_bbox_head.loss(data['gt_bboxes_3d'], data['gt_labels_3d'], outs, data['GGA_boxes_img'], data['GGA_lidar2img'],
                                      data['GGA_init_pseudo_labels'], data['GGA_bdry_masks'], data['GGA_in_box_points'], data['img_metas'], srl=srl)
        runs['sparse2'] = {k: float(v) for k, v in ls.items()}
    line = dict(case=case, seed=seed)
    for n, l in runs.items():
        worst = max(l64, key=lambda k: rel(l[k], l64[k]))
        line[n] = (round(rel(l[worst], l64[worst]), 9), worst)
    # the three largest-valued loss entries, to see how the synthetic case is conditioned
    line['largest'] = sorted(((round(v, 3), k) for k, v in l64.items()), reverse=True)[:3]
    rows.append(line)
    print(json.dumps(line), flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', f'precision_seeds_{case}.json'), 'w') as f:
    json.dump(rows, f, indent=1)
