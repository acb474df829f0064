"""Whole-step parity on the GPU: the product path (HIP kernels + MIOpen trunk) against the
oracle's CPU restatement with identical weights, inputs and SRL draws."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import REPO
from gga_amd import Config, build_model, synthetic
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PP_CFG = os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py')
GRAD_TOL = 1e-3      # per-parameter relative L2 error of the gradients against the float64 CPU restatement


def test_graft_smoke():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.parametrize('channels_last', [False, True])
def test_pp_train_step_matches_cpu_reference(channels_last):
    cfg = Config.fromfile(PP_CFG)
    if channels_last:
        cfg.model.pts_middle_encoder['channels_last'] = True
    torch.manual_seed(1)
    model = build_model(cfg.model)
    model.train()
    # Kaiming(fan_out) on the 2/3-channel output convs gives log-dims of std ~5 at init, i.e.
    # boxes of e^10 m: damp the regression outputs so the losses are O(1) and comparable
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    B = 2
    batch = synthetic.make_batch(B, start=50, n_points=5000, pc_range=synthetic.RANGE_PP, n_obj_range=(4, 8),
                                 n_ibp_range=(10, 200))
    ref = copy.deepcopy(model)
    ref64 = copy.deepcopy(model).double()         # the same step in float64: the yardstick for the gradients
    srl = model.pts_bbox_head.draw_srl(B)
    ref_losses, _ = R.reference_train_step(ref, batch, srl=srl)
    R.reference_train_step(ref64, batch, srl=srl)
    model.to(DEV)
    if channels_last:
        from gga_amd.cnn import to_channels_last
        to_channels_last(model)
    data = dict(batch, points=[p.to(DEV) for p in batch['points']])
    feats = model.extract_feat(data['points'], None, data['img_metas'])[1]
    outs = model.pts_bbox_head(feats)
    losses = model.pts_bbox_head.loss(data['gt_bboxes_3d'], data['gt_labels_3d'], outs, data['GGA_boxes_img'],
                                      data['GGA_lidar2img'], data['GGA_init_pseudo_labels'], data['GGA_bdry_masks'],
                                      data['GGA_in_box_points'], data['img_metas'], srl=srl)
    assert set(losses) == set(ref_losses) and len(losses) == 18
    for k, v in ref_losses.items():
        assert float(losses[k]) == pytest.approx(float(v), rel=1e-4, abs=1e-4), k     # north_star: within 1e-4
    total, log_vars = model._parse_losses(losses)
    total.backward()
    grads = {}
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        if p2.grad is None:
            assert p1.grad is None or float(p1.grad.abs().max()) == 0, n1
            continue
        grads[n1] = p1.grad.cpu()
    # every parameter's gradient: within GRAD_TOL of the float64 gradient, or - where fp32 itself
    # cannot get that close (early trunk layers, see oracle/torch_ref.gradient_offenders) - no
    # further from it than twice the fp32 CPU restatement is
    assert len(grads) > 100
    assert R.gradient_offenders(grads, ref, ref64, tol=GRAD_TOL, slack=2.0) == []


def test_loss_path_full_batch_vs_c_oracle():
    """The loss path at the bench size - 16 frames of 20 000 points, the 248 x 216 head maps - against the C
    oracle (``O.get_targets`` + ``O.head_loss``): every one of the 18 losses within 1e-4, and the gradient
    of the total w.r.t. the regression maps non-zero only on object cells."""
    from oracle import oracle as O
    cfg = Config.fromfile(PP_CFG)
    torch.manual_seed(0)
    head = build_model(cfg.model).pts_bbox_head
    B, T = 16, len(head.task_heads)
    tc = head.train_cfg
    batch = synthetic.make_batch(B, start=300, n_points=20000, pc_range=synthetic.RANGE_PP)
    fw, fh = (int(g) // int(tc['out_size_factor']) for g in tc['grid_size'][:2])
    preds = synthetic.make_head_preds(B, fh, fw, seed=91, n_tasks=T)
    srl = head.draw_srl(B)
    as_np = lambda xs: [a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a) for a in xs]
    tg = O.get_targets(as_np(batch['gt_labels_3d']), as_np(batch['GGA_boxes_img']), as_np(batch['GGA_lidar2img']),
                       as_np(batch['GGA_init_pseudo_labels']), as_np(batch['GGA_bdry_masks']),
                       [as_np(f) for f in batch['GGA_in_box_points']], [m['lidar2img'] for m in batch['img_metas']], tc,
                       np.asarray(srl, np.float32), n_tasks=T)
    want, _ = O.head_loss([{k: v.numpy() for k, v in p.items()} for p in preds], tg, tc)
    assert sum(int(tg['mask'][t].sum()) for t in range(T)) > 5 * B        # a loss over real objects
    head.to(DEV)
    maps = [{k: v.to(DEV).requires_grad_(True) for k, v in p.items()} for p in preds]
    losses = head.loss(batch['gt_bboxes_3d'], batch['gt_labels_3d'], [[m] for m in maps], batch['GGA_boxes_img'],
                       batch['GGA_lidar2img'], batch['GGA_init_pseudo_labels'], batch['GGA_bdry_masks'],
                       batch['GGA_in_box_points'], batch['img_metas'], srl=srl)
    assert set(losses) == set(want) and len(want) == 6 * T
    for k, v in want.items():
        assert float(losses[k]) == pytest.approx(float(v), rel=1e-4, abs=1e-4), k
    sum(losses.values()).backward()
    for t in range(T):
        cells = int(tg['mask'][t].sum())
        for k in ('reg', 'height', 'dim', 'rot'):
            g = maps[t][k].grad
            assert torch.isfinite(g).all()
            per_cell = (g != 0).any(dim=1).sum()
            assert 0 < int(per_cell) <= cells, (t, k)


def test_runner_steps_and_loss_decreases():
    from gga_amd.train import Runner
    cfg = Config.fromfile(PP_CFG)
    torch.manual_seed(0)
    model = build_model(cfg.model).to(DEV)
    with torch.no_grad():       # keep exp(log-dims) finite on noise inputs (see bench.damp_head_init)
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    runner = Runner(model, cfg, max_iters=100)
    b = synthetic.make_batch(2, n_points=4000, pc_range=synthetic.RANGE_PP)
    b['points'] = [p.to(DEV) for p in b['points']]
    data = {k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)}
    first = float(runner.step(data)['loss'])
    for _ in range(15):
        out = runner.step(data)
    last = float(out['loss'])
    assert np.isfinite(first) and np.isfinite(last) and last < first
    assert set(k for k in out['log_vars'] if k.startswith('task0.')) == {
        'task0.distancex', 'task0.distancey', 'task0.distancemin', 'task0.loss_heatmap', 'task0.loss_bbox',
        'task0.loss_ratio'}


def test_multi_step_trajectory_matches_cpu_restatement():
    """several optimizer steps of the GPU path against the CPU restatement with the
    same weights, batches, SRL draws (same torch seed), schedule, clipping and AdamW."""
    from gga_amd.train import Runner
    cfg = Config.fromfile(PP_CFG)
    torch.manual_seed(7)
    model = build_model(cfg.model)
    model.train()
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    ref = copy.deepcopy(model)
    B, n_steps = 2, 3
    batches = [synthetic.make_batch(B, start=70 + 10 * i, n_points=4000, pc_range=synthetic.RANGE_PP, n_obj_range=(4, 8),
                                    n_ibp_range=(10, 150)) for i in range(2)]
    # CPU restatement, none of it through the product: plain torch AdamW with the config's values, the oracle's own
    # restatement of mmcv's cyclic lr / momentum hooks, torch's gradient clipping, oracle/torch_ref's train step
    torch.manual_seed(123)
    oc = cfg.optimizer
    assert oc['type'] == 'AdamW' and 'paramwise_cfg' not in oc
    opt = torch.optim.AdamW(ref.parameters(), lr=oc['lr'], betas=tuple(oc['betas']), weight_decay=oc['weight_decay'])
    lr_s = lambda it: R.cyclic_value(oc['lr'], it, 100, (10, 1e-4), 1, 0.4)
    mo_s = lambda it: R.cyclic_value(oc['betas'][0], it, 100, (0.85 / 0.95, 1), 1, 0.4)
    ref_losses = []
    for it in range(n_steps):
        for g in opt.param_groups:
            g['lr'] = lr_s(it)
            g['betas'] = (mo_s(it), g['betas'][1])
        opt.zero_grad(set_to_none=True)
        _, total = R.reference_train_step(ref, batches[it % 2])
        torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.grad is not None], max_norm=35, norm_type=2)
        opt.step()
        ref_losses.append(float(total.detach()))
    # GPU path; before every step the CPU restatement is also evaluated from a copy of the GPU
    # model's CURRENT weights (same batch, same SRL draws): the loss curve along the real
    # trajectory, step by step, within the north-star tolerance
    torch.manual_seed(123)
    model.to(DEV)
    runner = Runner(model, cfg, max_iters=100)
    losses, resync = [], []
    for it in range(n_steps):
        b = batches[it % 2]
        data = {k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)}
        data['points'] = [p.to(DEV) for p in b['points']]
        here = copy.deepcopy(model).cpu()
        rng = torch.get_rng_state()
        _, total_here = R.reference_train_step(here, b, backward=False)
        torch.set_rng_state(rng)              # the GPU step draws the same SRL factors
        resync.append(float(total_here.detach()))
        losses.append(float(runner.step(data)['loss'].detach()))
    for it in range(n_steps):
        assert losses[it] == pytest.approx(resync[it], rel=1e-4), (it, losses, resync)
    # free-running trajectories (each side keeps its own weights): step 0 starts from identical
    # weights. AdamW divides by sqrt(v): where a gradient is ~0, fp32-rounding-level differences
    # between the two convolution implementations become lr-sized weight differences, so the
    # trajectories separate by about 10x per step
    for it, tol in enumerate((1e-4, 1e-3, 1e-2)):
        assert losses[it] == pytest.approx(ref_losses[it], rel=tol), (it, losses, ref_losses)
    num = sum(float((p.detach().cpu() - q.detach()).pow(2).sum()) for p, q in zip(model.parameters(), ref.parameters()))
    den = sum(float(q.detach().pow(2).sum()) for q in ref.parameters())
    assert (num / den) ** 0.5 < 1e-2          # weights after three AdamW steps (Adam normalises tiny gradient differences up)


def test_second_config_train_step_runs_and_learns():
    """configs/gga/gga_kitti_config.py (the reference's shipped model section: HardSimpleVFE +
    SparseEncoder + SECOND + SECONDFPN + CenterHead_GGA) end to end on the HIP path."""
    from gga_amd.train import Runner
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
    torch.manual_seed(0)
    model = build_model(cfg.model).to(DEV)
    with torch.no_grad():       # keep exp(log-dims) finite on noise inputs (see bench.damp_head_init)
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    runner = Runner(model, cfg, max_iters=100)
    b = synthetic.make_batch(2, n_points=20000, pc_range=synthetic.RANGE_SECOND)
    b['points'] = [p.to(DEV) for p in b['points']]
    data = {k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)}
    feats = model.extract_feat(data['points'], None, data['img_metas'])[1]
    assert feats[0].shape == (2, 512, 200, 176)
    first = float(runner.step(data)['loss'])
    for _ in range(10):
        out = runner.step(data)
    last = float(out['loss'])
    assert np.isfinite(first) and np.isfinite(last) and last < first
    assert len([k for k in out['log_vars'] if k.startswith('task')]) == 18


def test_second_config_prefetched_front_equals_inline_and_empty_batch():
    """Runner.step(data, next_data=...) runs voxelize + the sparse index plan of the next batch on a side
    stream; the step that consumes it must give exactly the losses of the in-line path. And a batch
    without a single point in range (zero sites on every level) steps without error."""
    import copy
    from gga_amd.train import Runner
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
    torch.manual_seed(0)
    model = build_model(cfg.model).to(DEV)
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    twin = copy.deepcopy(model)
    assert model.front_reads_counts
    batches = []
    for i in range(2):
        b = synthetic.make_batch(2, start=10 * i, n_points=8000, pc_range=synthetic.RANGE_SECOND)
        b['points'] = [p.to(DEV) for p in b['points']]
        batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
    ra, rb = Runner(model, cfg, max_iters=100), Runner(twin, cfg, max_iters=100)
    la, lb = [], []
    for i in range(4):
        torch.manual_seed(100 + i)          # same SRL draws on both sides
        la.append(float(ra.step(batches[i % 2])['loss']))
        torch.manual_seed(100 + i)
        lb.append(float(rb.step(batches[i % 2], next_data=batches[(i + 1) % 2])['loss']))
        assert i == 0 or len(rb._prepared) == 1
    # same kernels, same inputs, only the stream of the front differs. Step 0 is bit-identical; later steps
    # inherit the run-to-run noise of the few kernels that still sum with float atomics (the head loss's scatter of
    # gradients into shared cells), which AdamW amplifies step by step (one run in about twenty leaves 1e-5 at step 1)
    assert la[0] == lb[0] and la[1] == pytest.approx(lb[1], rel=1e-4), (la, lb)
    assert la[2:] == pytest.approx(lb[2:], rel=2e-2), (la, lb)
    # all points outside the range: zero voxels, zero sites on every level
    far = dict(batches[0], points=[torch.full((50, 4), 500.0, device=DEV) for _ in range(2)])
    out = ra.step(far)
    assert np.isfinite(float(out['loss']))


def _run_bench(extra, env=None, timeout=900):
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + extra, env=env, capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])


def test_bench_gpus2_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` as a plain command (no torchrun wrapper): the parent starts two
    ranks before touching the GPU (tools/dist_train.sh:10-20 in the reference). On a one-GPU box the
    ranks share the device over gloo; the real detector runs under DistributedDataParallel with the
    custom autograd Functions inside DDP's reducer hooks and host-side label inputs. Both configs:
    the PointPillars trunk as the main line, the shipped sparse trunk as `second_trunk`."""
    res = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2', '--second-batch', '2', '--pgd-batch', '1',
                      '--no-roofline'])
    assert res['n_gpus'] == 2 and res['value'] > 0 and res['config']['global_batch'] == 4
    assert res['config']['parallelism'] == 'dp2'
    n_dev = torch.cuda.device_count()
    assert res['config']['backend'].startswith('nccl' if n_dev >= 2 else 'gloo')
    st = res['second_trunk']
    assert st['global_batch'] == 4 and st['value'] > 0 and st['config_file'].endswith('gga_kitti_config.py')
    assert res['pgd_trunk']['global_batch'] == 2 and res['pgd_trunk']['value'] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one device per rank (>= 2 GPUs)')
def test_two_rank_rccl_both_configs():
    """Two ranks on two GPUs, backend nccl (= RCCL): gradient all-reduce over xGMI for the
    PointPillars config and for configs/gga/gga_kitti_config.py (BASELINE config #3's workload)."""
    env = dict(os.environ, GGA_DIST_BACKEND='nccl')
    res = _run_bench(['--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '4', '--second-batch', '4', '--pgd-batch', '2',
                      '--no-roofline'], env=env)
    assert res['n_gpus'] == 2 and res['config']['backend'].startswith('nccl')
    assert res['config']['global_batch'] == 8 and res['second_trunk']['global_batch'] == 8
    assert res['value'] > 0 and res['second_trunk']['value'] > 0


def test_bench_line_contract_single_gpu():
    """The default single-GPU line carries every field the driver and the judge read."""
    res = _run_bench(['--steps', '2', '--warmup', '1', '--batch', '2', '--second-batch', '2', '--pgd-batch', '1', '--no-cpu-baseline'])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'arith', 'data', 'config', 'roofline', 'mfma_roofline', 'second_trunk', 'pgd_trunk'):
        assert k in res, k
    assert res['roofline']['bound'] == 'hbm' and res['roofline']['launches_timed'] == 2
    assert res['mfma_roofline']['launches_timed'] == 2 * res['mfma_roofline']['launches_per_step']
    assert 0 < res['mfma_roofline']['share_of_step'] < 1
    assert 'fp16 planes' in res['arith'] and 'three f16 MFMA partial products' in res['arith'] and res['dtype'] == 'f32'


def test_edge_cases_empty_inputs():
    """Empty / degenerate inputs the reference's pipeline can produce: a batch whose frames carry no
    labelled object, a frame with no point inside the range, zero pillars, zero boxes."""
    from gga_amd import functional as F
    from gga_amd import ops
    cfg = Config.fromfile(PP_CFG)
    torch.manual_seed(0)
    model = build_model(cfg.model).to(DEV).train()
    b = synthetic.make_batch(2, n_points=3000, pc_range=synthetic.RANGE_PP, n_obj_range=(2, 3))
    # frame 0: every label ignored (-1); frame 1: no object at all
    b['gt_labels_3d'][0] = torch.full_like(b['gt_labels_3d'][0], -1)
    for k in ('gt_labels_3d', 'GGA_boxes_img', 'GGA_lidar2img', 'GGA_init_pseudo_labels', 'GGA_bdry_masks'):
        b[k][1] = b[k][1][:0]
    b['GGA_in_box_points'][1] = []
    b['points'] = [p.to(DEV) for p in b['points']]
    data = {k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)}
    out = model.train_step(data)
    out['loss'].backward()
    lv = {k: float(v) for k, v in out['log_vars'].items()}
    assert all(np.isfinite(v) for v in lv.values())
    assert all(lv[f'task{t}.loss_bbox'] == 0 and lv[f'task{t}.loss_ratio'] == 0 and lv[f'task{t}.distancemin'] == 0
               for t in range(3))
    assert all(lv[f'task{t}.loss_heatmap'] > 0 for t in range(3))        # pure background focal loss
    # a frame whose points all fall outside the range -> zero voxels for that frame
    far = torch.full((100, 4), 500.0, device=DEV)
    v, n, c, vn = F.hard_voxelize_batch([far, b['points'][0]], [0.16, 0.16, 4], synthetic.RANGE_PP, 32, 16000)
    assert vn.tolist()[0] == 0 and vn.tolist()[1] == len(c) and (c[:, 0] == 1).all()
    # zero pillars -> all-zero canvas (and the cell map stays clean for the next call)
    z = F.pillar_scatter(torch.zeros(0, 64, device=DEV), torch.zeros(0, 4, dtype=torch.int32, device=DEV), 2, 496, 432)
    assert z.shape == (2, 64, 496, 432) and float(z.abs().sum()) == 0
    # zero boxes / zero candidates
    dets, keep = ops.nms_rotated(torch.zeros(0, 5, device=DEV), torch.zeros(0, device=DEV), 0.3)
    assert keep.numel() == 0
    pib = ops.points_in_boxes_part(torch.zeros(1, 10, 3, device=DEV), torch.zeros(1, 0, 7, device=DEV))
    assert (pib == -1).all()
    assert ops.box_iou_rotated(torch.zeros(0, 5, device=DEV), torch.ones(3, 5, device=DEV)).shape == (0, 3)


def test_native_library_is_the_compute_path():
    """The process that ran the ops above must have the in-tree libgga_hip.so mapped."""
    from gga_amd import _lib
    _lib.lib()
    maps = open('/proc/self/maps').read()
    assert os.path.join(REPO, 'gga_amd', 'libgga_hip.so') in maps or 'libgga_hip.so' in maps
