"""Where does the whole-step deviation from float64 grow in the shipped config? Relative L2 error of the activations after
the sparse encoder, the backbone stages, the neck and every head map - GPU on two and on three planes, and the fp32 CPU
restatement - against the float64 CPU restatement (same weights, batch, SRL draws)."""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, synthetic, dense_conv
from gga_amd.cnn import to_channels_last
from oracle import torch_ref as R, sparse_ref as SR
DEV = 'cuda:0'
cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs', 'gga', 'gga_kitti_config.py'))
torch.manual_seed(3)
model = build_model(cfg.model)
model.train()
with torch.no_grad():
    for th in model.pts_bbox_head.task_heads:
        for name in ('reg', 'height', 'dim', 'rot'):
            getattr(th, name)[-1].weight.mul_(0.05)
B = 2
batch = synthetic.make_batch(B, start=50, n_points=20000, pc_range=synthetic.RANGE_SECOND, n_obj_range=(4, 8), n_ibp_range=(10, 200))
srl = model.pts_bbox_head.draw_srl(B)
torch.set_num_threads(min(os.cpu_count(), 32))


def hooked(m, store):
    hs = []
    def add(name, mod):
        hs.append(mod.register_forward_hook(lambda mod, inp, out, name=name: store.__setitem__(name, out)))
    add('backbone', m.pts_backbone)
    add('neck', m.pts_neck)
    add('shared_conv', m.pts_bbox_head.shared_conv)
    for t, th in enumerate(m.pts_bbox_head.task_heads):
        add(f'task{t}', th)
    return hs


def flat(store):
    out = {}
    for k, v in store.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out[f'{k}.{kk}'] = vv.detach().double().cpu()
        elif isinstance(v, (list, tuple)):
            for i, vv in enumerate(v):
                if isinstance(vv, dict):
                    for kk, vvv in vv.items():
                        out[f'{k}.{kk}'] = vvv.detach().double().cpu()
                else:
                    out[f'{k}.{i}'] = vv.detach().double().cpu()
        else:
            out[k] = v.detach().double().cpu()
    return out


def cpu_run(dtype):
    m = copy.deepcopy(model).to(dtype)
    store = {}
    hs = hooked(m, store)
    orig = SR.sparse_encoder_reference
    def enc(*a, **k):
        r = orig(*a, **k)
        store['sparse_encoder'] = r[0]
        return r
    SR.sparse_encoder_reference = enc
    losses, _ = R.reference_train_step(m, batch, srl=srl, backward=False)
    SR.sparse_encoder_reference = orig
    for h in hs:
        h.remove()
    return flat(store), {k: float(v) for k, v in losses.items()}


def gpu_run(planes):
    dense_conv.PLANES = planes
    m = copy.deepcopy(model)
    m.pts_middle_encoder.channels_last = True
    m = to_channels_last(m.to(DEV))
    store = {}
    hs = hooked(m, store)
    hs.append(m.pts_middle_encoder.register_forward_hook(lambda mod, inp, out: store.__setitem__('sparse_encoder', out)))
    data = dict(batch, points=[p.to(DEV) for p in batch['points']])
    feats = m.extract_feat(data['points'], None, data['img_metas'])[1]
    outs = m.pts_bbox_head(feats)
    losses = m.pts_bbox_head.loss(data['gt_bboxes_3d'], data['gt_labels_3d'], outs, data['GGA_boxes_img'], data['GGA_lidar2img'],
                                  data['GGA_init_pseudo_labels'], data['GGA_bdry_masks'], data['GGA_in_box_points'], data['img_metas'], srl=srl)
    return flat(store), {k: float(v) for k, v in losses.items()}


a64, l64 = cpu_run(torch.float64)
runs = {'cpu fp32': cpu_run(torch.float32), 'gpu 2 planes': gpu_run(2), 'gpu 3 planes': gpu_run(3)}
names = list(runs)
print(f'{"activation":28s} ' + ' '.join(f'{n:>14s}' for n in names))
for k in a64:
    print(f'{k:28s} ' + ' '.join(f'{float((runs[n][0][k] - a64[k]).norm() / a64[k].norm()):14.2e}' for n in names))
print('losses: relative deviation from float64')
for k in l64:
    if l64[k] == 0:
        continue
    print(f'{k:28s} ' + ' '.join(f'{abs(runs[n][1][k] - l64[k]) / abs(l64[k]):14.2e}' for n in names))
