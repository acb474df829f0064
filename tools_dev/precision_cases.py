"""The whole-step parity case of tests/test_model_gpu.py (_gpu_step_against: grad mode, the Runner's layouts) over several
seeds, batch sizes and object counts: worst deviation of the 18 losses from the float64 step and from the fp32 CPU
restatement, on two and on three planes. Usage: precision_cases.py second|pp B n_obj_lo n_obj_hi seed [seed ...]"""
import copy, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch
from gga_amd import Config, build_model, synthetic, dense_conv
from gga_amd.cnn import to_channels_last
from oracle import torch_ref as R
DEV = 'cuda:0'
case, B, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
seeds = [int(s) for s in sys.argv[5:]]
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'gga', 'gga_kitti_config.py' if case == 'second' else 'gga_kitti_pointpillars_config.py'))
torch.set_num_threads(min(os.cpu_count(), 32))
rel = lambda a, b: abs(a - b) / max(abs(b), 1.0)
rows = []
for seed in seeds:
    torch.manual_seed(seed)
    model = build_model(cfg.model)
    model.train()
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    kw = dict(n_points=20000, pc_range=synthetic.RANGE_SECOND) if case == 'second' else dict(n_points=5000, pc_range=synthetic.RANGE_PP)
    batch = synthetic.make_batch(B, start=50 + 10 * (seed - (3 if case == 'second' else 1)), n_obj_range=(lo, hi), n_ibp_range=(10, 200), **kw)
    srl = model.pts_bbox_head.draw_srl(B)
    l32, _ = R.reference_train_step(copy.deepcopy(model), batch, srl=srl, backward=False)
    l64, _ = R.reference_train_step(copy.deepcopy(model).double(), batch, srl=srl, backward=False)
    l32, l64 = {k: float(v) for k, v in l32.items()}, {k: float(v) for k, v in l64.items()}
    line = dict(case=case, B=B, n_obj=(lo, hi), seed=seed, objects=sum(len(x) for x in batch['gt_labels_3d']),
                cpu32=max(rel(l32[k], l64[k]) for k in l64))
    for planes in (2, 3):
        dense_conv.PLANES = planes
        m = copy.deepcopy(model)
        m.pts_middle_encoder.channels_last = True
        m = to_channels_last(m.to(DEV))
        data = dict(batch, points=[p.to(DEV) for p in batch['points']])
        feats = m.extract_feat(data['points'], None, data['img_metas'])[1]
        outs = m.pts_bbox_head(feats)
        losses = m.pts_bbox_head.loss(data['gt_bboxes_3d'], data['gt_labels_3d'], outs, data['GGA_boxes_img'], data['GGA_lidar2img'],
                                      data['GGA_init_pseudo_labels'], data['GGA_bdry_masks'], data['GGA_in_box_points'], data['img_metas'], srl=srl)
        l = {k: float(v.detach()) for k, v in losses.items()}
        w = max(l64, key=lambda k: rel(l[k], l64[k]))
        line[f'gpu{planes}_vs_f64'] = (rel(l[w], l64[w]), w)
        w = max(l32, key=lambda k: rel(l[k], l32[k]))
        line[f'gpu{planes}_vs_cpu32'] = (rel(l[w], l32[w]), w)
        del m, feats, outs, losses
    rows.append(line)
    print(json.dumps(line), flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', f'precision_cases_{case}_B{B}_{lo}_{hi}.json'), 'w') as f:
    json.dump(rows, f, indent=1)
