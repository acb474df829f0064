"""How far apart are two PGD steps of the same inputs: the same mode twice (float atomics), and the scheduling modes of
test_level_streams_and_prepared_targets_change_nothing against the plain one. Per parameter: max |d| / max |g| and |d|_2 / |g|_2."""
import os, sys, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, synthetic, mono3d_heads
from gga_amd.cnn import to_channels_last
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
DEV = 'cuda:0'
cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_pdg.py'))
torch.manual_seed(0)
model = to_channels_last(build_model(cfg.model).to(DEV))
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model.init_weights()
synthetic.damp_random_backbone(model)
model.train()
b = synthetic.make_mono_batch(2, device=DEV, img_hw=(192, 640))
data = {k: b[k] for k in synthetic.MONO_BATCH_KEYS}
data['img'] = data['img'].contiguous(memory_format=torch.channels_last)
state = {k: v.clone() for k, v in model.state_dict().items()}


def one(mode, prepared):
    mono3d_heads.LEVEL_BATCH, mono3d_heads.LEVEL_STREAMS = mode == 'batch', mode == 'streams'
    type(model.bbox_head).prepare_loss_before_forward = prepared
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    out = model.train_step(data)
    out['loss'].backward()
    torch.cuda.synchronize()
    return {k: float(v) for k, v in out['log_vars'].items()}, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


def cmp(a, b, name):
    la, ga = a; lb, gb = b
    dl = max(abs(la[k] - lb[k]) / (abs(lb[k]) + 1e-12) for k in la)
    mx = sorted(((float((ga[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-30)), n) for n in ga), reverse=True)
    l2 = sorted(((float((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-30)), n) for n in ga), reverse=True)
    tot = float(torch.sqrt(sum((ga[n] - gb[n]).double().pow(2).sum() for n in ga)) / torch.sqrt(sum(gb[n].double().pow(2).sum() for n in gb)))
    print(f'{name}: worst rel loss diff {dl:.2e}; max-rule worst {mx[0][0]:.2e} ({mx[0][1]}), {sum(m > 2e-3 for m, _ in mx)} over 2e-3; '
          f'L2 rule worst {l2[0][0]:.2e} ({l2[0][1]}); whole gradient {tot:.2e}')


p0 = one('plain', False)
cmp(one('plain', False), p0, 'plain twice      ')
cmp(one('plain', False), p0, 'plain thrice     ')
cmp(one('batch', True), p0, 'batch + prepared ')
cmp(one('streams', True), p0, 'streams + prepared')
cmp(one('streams', True), p0, 'streams again    ')
