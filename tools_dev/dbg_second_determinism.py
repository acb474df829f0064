"""Is the shipped config's step bit-reproducible? Two Runners from the same seed on the same resident batches (with / without the
prefetched front), losses of 12 steps as hex."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gga_amd import Config, build_model, synthetic
from gga_amd.cnn import to_channels_last
from gga_amd.train import Runner
case = sys.argv[1] if len(sys.argv) > 1 else 'second'
cfgf, rng = (('gga_kitti_config.py', synthetic.RANGE_SECOND) if case == 'second' else ('gga_kitti_pointpillars_config.py', synthetic.RANGE_PP))
prefetch = (sys.argv[2] if len(sys.argv) > 2 else '1') == '1'
dev = torch.device('cuda:0')


def run():
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', cfgf))
    cfg.model.pts_middle_encoder['channels_last'] = True
    torch.manual_seed(3)
    model = build_model(cfg.model)
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    model = to_channels_last(model.to(dev)).train()
    runner = Runner(model, cfg, max_iters=100)
    batches = []
    for i in range(2):
        b = synthetic.make_batch(2, start=10 * i, n_points=20000, pc_range=rng)
        b['points'] = [p.to(dev) for p in b['points']]
        batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
    torch.cuda.synchronize()
    runner.inputs_ready(*batches)
    torch.manual_seed(5)
    out = []
    for i in range(12):
        o = runner.step(batches[i % 2], next_data=batches[(i + 1) % 2] if prefetch else None)
        out.append(float(o['loss'].detach()).hex())
    return out
a, b = run(), run()
print(case, 'prefetch', prefetch, 'identical' if a == b else 'DIFFERENT', [i for i, (x, y) in enumerate(zip(a, b)) if x != y])
print(a[:4], b[:4])
