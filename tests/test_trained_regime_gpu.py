"""Whole-step parity in the TRAINED regime (VERDICT r05 item 1 / weak #1b-c).

Every other whole-step case starts from Kaiming initialisation on uniformly scattered points. Block-scaled 16-bit operand
planes are the arithmetic whose error depends on the operands' distribution, so this case looks where a run actually lives:

* the REAL job - ``train.train_detector`` on a synthetic KITTI tree on disk (``synthetic.write_kitti_tree``), the reference's
  full train pipeline of configs/gga/gga_kitti_config.py:93-137 incl. ``ObjectSample_GGA`` database sampling in loader
  workers, group sampler, collate, upload, ``Runner.step`` with the config's AdamW / cyclic schedules / clipping - for
  ``STEPS`` >= 300 optimizer steps (mmdet3d/apis/train.py:180-322);
* every 50th step (PointPillars; every 100th on the shipped config) is RE-SYNCED: before the step the CPU restatement (oracle/torch_ref.reference_train_step, fp32)
  is evaluated from a copy of the GPU model's CURRENT weights on the batch the loader just delivered, with the SRL draws the
  step is about to make, and all 18 losses of the GPU step must lie within 1e-4 of it - the "loss curve within 1e-4" claim
  of BASELINE.json's north_star, along the real trajectory (centerpoint_head_gga.py:629-723);
* after the last epoch ONE more step from the trained weights is compared with the restatement in fp32 (plain 1e-4 on all
  18 losses) AND in float64 (1e-4, or twice the fp32 CPU step's own distance where fp32 itself is further: the clamped heat
  map, see below), for both arithmetic forms (two fp16 planes = what the Runner ran, three bf16 planes), and the
  per-parameter gradients against float64 (oracle/torch_ref.gradient_offenders);
* the range guard runs every ``EVERY`` steps and its statistics are printed: the run must stay on two planes.
"""
import copy
import json
import os
import tempfile

import pytest
import torch

from conftest import REPO
from gga_amd import Config, build_model, synthetic
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
STEPS, EVERY = 300, 50
SEED = {'pp': 11, 'second': 11}
RESYNC_EVERY = {'pp': 50, 'second': 100}        # the shipped config's CPU step takes ~12 s: three re-synced steps there, six on PointPillars
CASES = {
    # name: (config, point-cloud range, frames on disk, frames per step)
    'pp': ('gga_kitti_pointpillars_config.py', synthetic.RANGE_PP, 64, 2),
    'second': ('gga_kitti_config.py', synthetic.RANGE_SECOND, 64, 2),
}
rel = lambda a, b: abs(a - b) / max(abs(b), 1.0)          # absolute 1e-4 below 1, relative above (as tests/test_model_gpu.py)


def _cpu_twin(cfg, gpu_model, dtype=torch.float32):
    """A CPU model with the GPU model's current parameters and buffers (built fresh from the config: no device-side state)."""
    twin = build_model(cfg.model)
    twin.load_state_dict({k: v.detach().cpu() for k, v in gpu_model.state_dict().items()})
    return twin.to(dtype).train()


def _cpu_batch(data):
    out = dict(data)
    out['points'] = [p.detach().cpu() for p in data['points']]
    return out


@pytest.mark.parametrize('case', ['pp', 'second'])
def test_trained_regime_parity(case, monkeypatch):
    # In a process of its own: at the end of a 25-minute pytest session (hundreds of GPU tests behind it, host thread pools
    # limited and re-opened by earlier cases, tens of GB touched) this case ran 5 x slower than alone (480 / 550 s against
    # 98 / 160 s: its CPU restatements). The child runs this same function with GGA_TRAINED_REGIME_INNER set.
    if not os.environ.get('GGA_TRAINED_REGIME_INNER'):
        import subprocess
        import sys
        out = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-s', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                              '-k', f'test_trained_regime_parity and {case}'], env=dict(os.environ, GGA_TRAINED_REGIME_INNER='1'),
                             capture_output=True, text=True, timeout=1500, cwd=REPO)
        print('\n'.join(l for l in out.stdout.splitlines() if l.startswith(('RESYNC', 'TRAINED', 'GUARD'))))
        assert out.returncode == 0, (out.stdout[-4000:], out.stderr[-2000:])
        assert f'TRAINED_STEP {case} planes 2' in out.stdout and f'TRAINED_STEP {case} planes 3' in out.stdout
        return
    from gga_amd import dense_conv
    from gga_amd.cnn import to_channels_last
    from gga_amd.loader import build_dataset
    from gga_amd.train import Runner, setup_multi_processes, train_detector
    cfg_name, rng, frames, B = CASES[case]
    root = os.path.join(tempfile.gettempdir(), f'gga_trained_regime_{case}_{os.getuid()}')
    info_path, db_path = synthetic.write_kitti_tree(root, frames, pc_range=rng, db_per_class=100)
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', cfg_name))
    d = cfg.data['train']
    d['dataset'].update(data_root=root + '/', ann_file=info_path)
    for t in d['dataset']['pipeline']:
        if t['type'] == 'ObjectSample_GGA':
            t['db_sampler'].update(data_root=root + '/', info_path=db_path)
        if 'point_cloud_range' in t:
            t['point_cloud_range'] = list(rng)
    iters_per_epoch = frames // B
    epochs = -(-STEPS // iters_per_epoch)
    # (persistent workers: the loader forks its two workers once, not once per epoch)
    cfg.data.update(samples_per_gpu=B, workers_per_gpu=2, persistent_workers=True)
    cfg.runner = dict(type='EpochBasedRunner', max_epochs=epochs)
    cfg.checkpoint_config, cfg.work_dir, cfg.seed = None, None, 0
    cfg['gga_range_check_interval'] = EVERY
    setup_multi_processes(cfg)
    cfg.model.pts_middle_encoder['channels_last'] = True
    monkeypatch.setattr(dense_conv, 'PLANES_PINNED', False)
    # the reference's set_random_seed (tools/train.py:214-219; tools/train.py here): python, numpy, torch - the database sampler draws its
    # order from numpy's global generator when the dataset is built, so without it every run trains on other frames
    from gga_amd.train import set_random_seed
    set_random_seed(SEED[case])
    model = build_model(cfg.model)
    with torch.no_grad():       # random init only: keep exp(log-dims) finite on noise (bench.damp_head_init)
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    model = to_channels_last(model.to(DEV)).train()
    dataset = build_dataset(d)

    threads = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    curve, last, trail = [], {}, []
    real_step = Runner.step

    def step(self, data, next_data=None):
        check = self.iter > 0 and self.iter % RESYNC_EVERY[case] == 0
        if check:
            state = torch.get_rng_state()                # (before the twin is built: its weight initialisation draws from the generator too)
            twin = _cpu_twin(cfg, self.raw_model)
            torch.set_rng_state(state)
            ref, _ = R.reference_train_step(twin, _cpu_batch(data), backward=False)
            torch.set_rng_state(state)                   # the step below draws the same SRL factors
        out = real_step(self, data, next_data)
        if check:
            got = {k: float(v) for k, v in out['log_vars'].items() if k in ref}
            worst = max(ref, key=lambda k: rel(got[k], float(ref[k])))
            curve.append(dict(iter=self.iter - 1, planes=self.planes, loss=float(out['loss'].detach()),
                              worst=rel(got[worst], float(ref[worst])), key=worst))
            print(f'RESYNC {case} iter {self.iter - 1} planes {self.planes} total {curve[-1]["loss"]:.4f}: worst of 18 losses vs the '
                  f'fp32 CPU step from the same weights {curve[-1]["worst"]:.2e} ({worst})')
            for k, v in ref.items():
                assert rel(got[k], float(v)) <= 1e-4, (self.iter - 1, k, got[k], float(v))
        if self.iter % 20 == 0:
            trail.append(round(float(out['loss'].detach()), 2))
        last['data'] = data
        return out
    monkeypatch.setattr(Runner, 'step', step)
    try:
        runner = train_detector(model, dataset, cfg, distributed=False, device=torch.device(DEV))
        monkeypatch.setattr(Runner, 'step', real_step)
        assert runner.iter >= STEPS and len(curve) >= STEPS // RESYNC_EVERY[case]
        assert runner.planes == 2 and not runner.fell_back, runner.range_reports       # the guard saw nothing to fall back for
        print('GUARD ' + case + ' ' + json.dumps([{k: r[k] for k in ('iter', 'operands', 'over_limit', 'forward_worst_share_lost',
                                                                      'backward_worst_mass_lost', 'worst_share_below_2p17')}
                                                   for r in runner.range_reports]))
        assert len(runner.range_reports) >= STEPS // EVERY
        # the loss moved: this is not the initialisation any more
        first_total = curve[0]['loss']
        print(f'TRAINED {case}: {runner.iter} optimizer steps, total loss at the re-synced steps ' + ' '.join(f'{c["loss"]:.3f}' for c in curve))
        print(f'TRAINED_CURVE {case} (every 20th step): {trail}')

        # ---- one step from the TRAINED weights: float64 and fp32 restatements, both arithmetic forms
        data = last['data']
        batch = _cpu_batch(data)
        ref32, ref64 = _cpu_twin(cfg, model), _cpu_twin(cfg, model, torch.float64)
        srl = model.pts_bbox_head.draw_srl(B)
        l32, _ = R.reference_train_step(ref32, batch, srl=srl)
        l64, _ = R.reference_train_step(ref64, batch, srl=srl)
        l32, l64 = {k: float(v) for k, v in l32.items()}, {k: float(v) for k, v in l64.items()}
        floor = max(rel(l32[k], l64[k]) for k in l64)
        for planes in (2, 3):
            monkeypatch.setattr(dense_conv, 'PLANES', planes)
            dense_conv.AMAX_POOL.next_generation()
            model.zero_grad(set_to_none=True)
            feats = model.extract_feat(data['points'], None, data['img_metas'])[1]
            outs = model.pts_bbox_head(feats)
            losses = model.pts_bbox_head.loss(data['gt_bboxes_3d'], data['gt_labels_3d'], outs, data['GGA_boxes_img'],
                                              data['GGA_lidar2img'], data['GGA_init_pseudo_labels'], data['GGA_bdry_masks'],
                                              data['GGA_in_box_points'], data['img_metas'], srl=srl)
            got = {k: float(v) for k, v in losses.items()}
            w64 = max(l64, key=lambda k: rel(got[k], l64[k]))
            w32 = max(l32, key=lambda k: rel(got[k], l32[k]))
            print(f'TRAINED_STEP {case} planes {planes} after {runner.iter} steps: worst deviation from float64 {rel(got[w64], l64[w64]):.2e} ({w64}), '
                  f'from the fp32 CPU step {rel(got[w32], l32[w32]):.2e} ({w32}); fp32 CPU step from float64 {floor:.2e}; bound 1e-4')
            # The reference's path is fp32 (docker/Dockerfile:1-9: PyTorch 1.6, no fp16 hook in the config): within 1e-4 of the
            # fp32 CPU step. Against float64: 1e-4, or - where fp32 arithmetic itself cannot get that close - no further than
            # twice the fp32 CPU step is on that key. It cannot on a TRAINED heat map: most cells sit at the clamp p = 1e-4, and
            # log(1 - p) in fp32 rounds 1 - 1e-4 to 0.99989998, the same 1.7e-8 for every clamped cell - a systematic 1.3e-4 of
            # loss_heatmap for GPU and CPU fp32 alike (measured round 6: both 1.33e-4 from float64, 1.7e-6 from each other).
            for k in l64:
                assert rel(got[k], l32[k]) <= 1e-4, (planes, k, got[k], l32[k])
                assert rel(got[k], l64[k]) <= max(1e-4, 2.0 * rel(l32[k], l64[k])), (planes, k, got[k], l64[k], l32[k])
            total, _ = model._parse_losses(losses)
            total.backward()
            grads = {n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None}
            assert len(grads) > 100
            strict = R.gradient_offenders(grads, ref32, ref64, tol=1e-3, slack=2.0)
            print(f'TRAINED_GRADS {case} planes {planes}: {len(grads)} parameters, over 1e-3 / twice the fp32 floor: '
                  f'{[(n, round(e, 5), round(f, 6)) for n, e, f in strict]}')
            # (two planes: at most 1 % of the parameters beyond the strict criterion, each within 1e-2 - what a flipped ReLU decision at
            # one of the few object cells of a head branch or a BatchNorm bias of the trunk costs; tests/test_model_gpu.py has the case)
            assert len(strict) <= (0.01 if planes == 2 else 0.05) * len(grads) and all(e <= 1e-2 for _, e, _ in strict), strict
    finally:
        monkeypatch.setattr(Runner, 'step', real_step)
        torch.set_num_threads(threads)
