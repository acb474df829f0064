"""Pure-PyTorch fp32 restatement of the floating-point half of the GGA train step, with
autograd. TEST INFRASTRUCTURE ONLY (see oracle/gga_oracle.c header): used by ``tests/`` as
the gradient / whole-step checker for the HIP path and by ``bench.py``'s ``cpu_baseline``
leg; the product never imports it.

It restates, op for op but batched instead of looped, the reference's
``CenterHead_GGA.loss`` (mmdet3d/models/dense_heads/centerpoint_head_gga.py:629-723),
``get_prediction_single`` (:250-341), ``get_distance_single`` (:184-239),
``PointPillarsScatter.forward_batch`` (middle_encoders/pillar_scatter.py:62-102) and mmdet's
``GaussianFocalLoss`` / ``L1Loss``; ``reference_train_step`` is the whole step of
``GGA.forward_train`` (detectors/centerpoint_gga.py:43-86) for both middle encoders - the pillar scatter
and the SparseEncoder of the shipped config (through oracle/sparse_ref.py). tests/test_torch_ref.py pins it to the golden vectors of
the imported reference (losses AND gradients).
"""
import numpy as np
import torch

from . import oracle as O

EPS32 = torch.finfo(torch.float32).eps


def clip_sigmoid(x, eps=1e-4):
    return torch.clamp(torch.sigmoid(x), min=eps, max=1 - eps)


def gaussian_focal(pred, target, alpha, gamma, avg_factor):
    eps = 1e-12
    pos = target.eq(1)
    neg_w = (1 - target).pow(gamma)
    pos_loss = -(pred + eps).log() * (1 - pred).pow(alpha) * pos
    neg_loss = -(1 - pred + eps).log() * pred.pow(alpha) * neg_w
    return (pos_loss + neg_loss).sum() / (avg_factor + EPS32)


def l1(pred, target, weight, avg_factor, loss_weight):
    return loss_weight * ((torch.abs(pred - target) * weight).sum() / (avg_factor + EPS32))


def box_geometry(pred, ind, lidar2img, tc):
    """pred [B,K,8] -> rot [B,K], ratio [B,K,2], box2d [B,K,4], bev [B,K,5]."""
    fw = int(tc['grid_size'][0]) // int(tc['out_size_factor'])
    vs = torch.tensor(tc['voxel_size'], dtype=torch.float32)
    pc = torch.tensor(tc['point_cloud_range'], dtype=torch.float32)
    osf = tc['out_size_factor']
    rot = torch.atan2(pred[..., 6], pred[..., 7])
    X = ((ind % fw) + pred[..., 0]) * vs[0] * osf + pc[0]
    Y = (torch.div(ind, fw, rounding_mode='trunc') + pred[..., 1]) * vs[1] * osf + pc[1]
    dims = torch.exp(pred[..., 3:6])
    zb = pred[..., 2] - dims[..., 2] * 0.5
    ox = torch.tensor([-.5, -.5, -.5, -.5, .5, .5, .5, .5])
    oy = torch.tensor([-.5, -.5, .5, .5, -.5, -.5, .5, .5])
    oz = torch.tensor([0., 1., 1., 0., 0., 1., 1., 0.])
    lx, ly, lz = dims[..., 0:1] * ox, dims[..., 1:2] * oy, dims[..., 2:3] * oz
    c, s = torch.cos(rot)[..., None], torch.sin(rot)[..., None]
    x = lx * c - ly * s + X[..., None]
    y = lx * s + ly * c + Y[..., None]
    z = lz + zb[..., None]
    M = lidar2img
    q = [M[..., i, 0:1] * x + M[..., i, 1:2] * y + M[..., i, 2:3] * z + M[..., i, 3:4] for i in range(3)]
    depth = torch.maximum(q[2], torch.tensor(0.1))
    u, v = q[0] / depth, q[1] / depth
    box = torch.stack([u.min(-1)[0], v.min(-1)[0], u.max(-1)[0], v.max(-1)[0]], -1)
    bev = torch.stack([X, Y, dims[..., 0], dims[..., 1], rot], -1)
    return rot, dims[..., :2], box, bev


def pal_distances(ibp, bev):
    """ibp: per-frame lists of [Ni,>=2] arrays for one task; bev [B,K,5] -> three [B,K,1]."""
    B, K, _ = bev.shape
    outs = [[torch.zeros(K) for _ in range(B)] for _ in range(3)]
    for b in range(B):
        for k, pts in enumerate(ibp[b]):
            p = torch.as_tensor(np.asarray(pts)[:, :2]).float()
            bx = bev[b, k]
            c, s = torch.cos(bx[4]), torch.sin(bx[4])
            rx, ry = p[:, 0] * c + p[:, 1] * s, -p[:, 0] * s + p[:, 1] * c
            cx, cy = bx[0] * c + bx[1] * s, -bx[0] * s + bx[1] * c
            hl, hw = bx[2] / 2.0, bx[3] / 2.0
            d = torch.stack([rx - (cx - hl), rx - (cx + hl), ry - (cy - hw), ry - (cy + hw)], 1).abs()
            vals = (d.min(1)[0].sum(), torch.relu((rx - cx).abs() - 2 * hl).sum(),
                    torch.relu((ry - cy).abs() - 2 * hw).sum())
            for j in range(3):
                m = torch.zeros(K)
                m[k] = 1.0
                outs[j][b] = outs[j][b] + m * vals[j]
    return [torch.stack(o)[..., None] for o in outs]


def head_loss(preds, tg, tc, alpha=0.0, gamma=4.0, l1_weight=0.25):
    """preds[t]: dict of NCHW tensors (heatmap = raw logits); tg: oracle.get_targets output."""
    losses = {}
    cw = torch.tensor(tc['code_weights'], dtype=torch.float32)
    for t, pd in enumerate(preds):
        hm_t = torch.from_numpy(tg['heatmap'][t])
        num_pos = float(hm_t.eq(1).float().sum())
        lh = gaussian_focal(clip_sigmoid(pd['heatmap']), hm_t, alpha, gamma, max(num_pos, 1))
        ind = torch.from_numpy(tg['ind'][t])
        msk = torch.from_numpy(tg['mask'][t])
        anno = torch.from_numpy(tg['anno_box'][t])
        cat = torch.cat([pd['reg'], pd['height'], pd['dim'], pd['rot']], 1)
        B, C = cat.shape[:2]
        pred = cat.view(B, C, -1).gather(2, ind[:, None, :].expand(B, C, ind.shape[1])).transpose(1, 2)
        rot, ratio, box, bev = box_geometry(pred, ind, torch.from_numpy(tg['lidar2img'][t]), tc)
        num = msk.float().sum()
        avg = num + 1e-4
        bw = msk[..., None].float() * (~torch.isnan(anno)).float() * cw
        dmin, dx, dy = pal_distances(tg['ibp'][t], bev)
        zero = torch.zeros_like(dmin)
        losses[f'task{t}.distancex'] = l1(dx, zero, bw[..., 0:1], avg, l1_weight) * 0.1
        losses[f'task{t}.distancey'] = l1(dy, zero, bw[..., 0:1], avg, l1_weight) * 0.1
        losses[f'task{t}.distancemin'] = l1(dmin, zero, bw[..., 0:1], avg, l1_weight) * 0.1
        rw, rl = ratio.min(-1, keepdim=True)[0], ratio.max(-1, keepdim=True)[0]
        srl = rl - rw * anno[..., 4:5]
        losses[f'task{t}.loss_heatmap'] = lh * 5.0
        wb = bw[..., :4] * torch.from_numpy(tg['bound_mask'][t]).float()
        losses[f'task{t}.loss_bbox'] = l1(box, anno[..., :4], wb, avg, l1_weight) * 0.3
        losses[f'task{t}.loss_ratio'] = l1(srl, torch.zeros_like(srl), bw[..., 4:5], avg, l1_weight) * 0.1
    return losses


def scatter(feats, coors, batch_size, ny, nx):
    C = feats.shape[1]
    canvas = feats.new_zeros(batch_size, C, ny * nx)
    idx = coors[:, 2].long() * nx + coors[:, 3].long()
    canvas[coors[:, 0].long(), :, idx] = feats
    return canvas.view(batch_size, C, ny, nx)


def reference_train_step(model, batch, srl=None, backward=True):
    """Whole GGA train step on the CPU for a model built by ``gga_amd.build_model`` (CPU
    parameters): the HIP-backed stages are replaced by the C oracle (voxelize) and the torch
    restatements above; the dense trunk / head convs are the model's own ``nn`` modules.
    Returns (loss dict, total)."""
    head = model.pts_bbox_head
    tc = head.train_cfg
    vl = model.pts_voxel_layer
    pts = [np.ascontiguousarray(p.detach().cpu().numpy(), np.float32) for p in batch['points']]
    v, n, c = O.voxelize_batch(pts, vl.voxel_size, vl.point_cloud_range, vl.max_num_points,
                               vl.max_voxels[0] if model.training else vl.max_voxels[1])
    v, n, c = torch.from_numpy(v), torch.from_numpy(n), torch.from_numpy(c)
    dtype = next(model.parameters()).dtype       # float64 models: the same step as a float64 yardstick
    v = v.to(dtype)
    enc = model.pts_voxel_encoder
    if type(enc).__name__ == 'HardSimpleVFE':
        feats = torch.from_numpy(O.voxel_mean(v.float().numpy(), n.numpy(), enc.num_features)).to(dtype)
    else:
        feats = enc(v, n, c)                      # PillarFeatureNet: plain torch modules
    me = model.pts_middle_encoder
    if hasattr(me, 'conv_input'):
        # SparseEncoder (centerpoint_gga.py:49-51 -> sparse_encoder.py:107-138): the pair-list restatement of every
        # sparse convolution (oracle/sparse_ref.py), BatchNorm1d / ReLU = the encoder's own torch modules; dispatch is
        # on attributes, nothing of libgga_hip runs
        from . import sparse_ref
        x, _ = sparse_ref.sparse_encoder_reference(me, feats, c, len(pts), pairs=True)
    else:
        x = scatter(feats, c, len(pts), me.ny, me.nx)
    x = model.pts_backbone(x)
    x = model.pts_neck(x)
    outs = head(x)
    B = len(pts)
    if srl is None:
        srl = O.draw_srl(B, len(head.task_heads))
    as_np = lambda xs: [a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a) for a in xs]
    tg = O.get_targets(as_np(batch['gt_labels_3d']), as_np(batch['GGA_boxes_img']), as_np(batch['GGA_lidar2img']),
                       as_np(batch['GGA_init_pseudo_labels']), as_np(batch['GGA_bdry_masks']),
                       [as_np(f) for f in batch['GGA_in_box_points']],
                       [m['lidar2img'] for m in batch['img_metas']], tc, srl, n_tasks=len(head.task_heads))
    preds = [o[0] for o in outs]
    losses = head_loss(preds, tg, tc, alpha=head.loss_cls.alpha, gamma=head.loss_cls.gamma,
                       l1_weight=head.loss_bbox.loss_weight)
    total = sum(v for k, v in losses.items() if 'loss' in k or (head.pal_backprop and 'distance' in k))
    if backward:
        total.backward()
    return losses, total


def gradient_offenders(named_grads, ref32, ref64, tol=1e-3, slack=2.0):
    """Per-parameter gradient check with a float64 yardstick. ``named_grads``: {name: grad (CPU
    tensor)} of the path under test; ``ref32`` / ``ref64``: the same model after
    ``reference_train_step`` in float32 / float64 on the same batch. A parameter offends when its
    relative L2 error against the float64 gradient exceeds ``tol`` AND ``slack`` times the error
    the float32 CPU restatement itself has against float64 - fp32 rounding through ~40 conv + BN
    layers is amplified to ~1e-2 in the early trunk layers on small batches (measured:
    tools_dev/grad_errors.py), for the CPU restatement and the GPU path alike, so 'within 1e-3 of
    another fp32 implementation' is not a meaningful bound there; 'no further from the truth than
    fp32 itself' is. -> list of (name, err_vs_f64, fp32_noise_floor)."""
    p32, p64 = dict(ref32.named_parameters()), dict(ref64.named_parameters())
    bad = []
    for name, g in named_grads.items():
        g64 = p64[name].grad
        if g64 is None or float(g64.norm()) < 1e-9:
            continue
        d = float(g64.norm())
        err = float((g.double() - g64).norm()) / d
        floor = float((p32[name].grad.double() - g64).norm()) / d
        if err > max(tol, slack * floor):
            bad.append((name, err, floor))
    return bad


# ----------------------------------------------------------------------------- optimizer schedule (a15)
def cyclic_value(base, it, max_iters, target_ratio, cyclic_times=1, step_ratio_up=0.4):
    """mmcv 1.x ``CyclicLrUpdaterHook`` / ``CyclicMomentumUpdaterHook`` by iteration (mmcv/runner/hooks/lr_updater.py,
    momentum_updater.py; the wheel is un-vendored - restated from the published algorithm, parity unpinned):
    two cosine phases per cycle, base -> base*target_ratio[0] over the first ``step_ratio_up`` of the cycle, then
    -> base*target_ratio[1]. Independent of gga_amd.train.CyclicSchedule (the tests compare the two)."""
    import math
    per_cycle = max_iters // cyclic_times
    up_end = int(step_ratio_up * per_cycle)
    cur = it % per_cycle
    if cur < up_end:
        start, end, frac = base, base * target_ratio[0], cur / up_end
    else:
        start, end, frac = base * target_ratio[0], base * target_ratio[1], (cur - up_end) / (per_cycle - up_end)
    return end + 0.5 * (start - end) * (math.cos(math.pi * frac) + 1.0)


def step_value(base, it, steps, gamma=0.1, warmup_iters=0, warmup_ratio=0.1):
    """mmcv ``StepLrUpdaterHook`` with linear warm-up by iteration (lr_updater.py: regular lr = base * gamma^k with k
    the number of milestones passed; during warm-up multiplied by 1 - (1 - it/warmup_iters)(1 - warmup_ratio))."""
    lr = base * gamma ** sum(1 for m in steps if it >= m)
    if it < warmup_iters:
        lr *= 1.0 - (1.0 - it / warmup_iters) * (1.0 - warmup_ratio)
    return lr
