"""Parity of the gfx950 kernels (through the C ABI) against the CPU oracle and the
golden vectors of the imported reference. Run with ``-m gpu`` on an MI355X.

Bars (BASELINE.json north_star): voxel / scatter indices and payload copies bit-exact;
fp32 losses within 1e-4; gradients against the reference's autograd output.
"""
import numpy as np
import pytest
import torch

from conftest import FMAP, TRAIN_CFG, load_head_case
from gga_amd import functional as F
from gga_amd import synthetic
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _vox_gpu(points_list, vs, rng, mp, mv):
    pts = [torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(DEV) for p in points_list]
    v, n, c, vn = F.hard_voxelize_batch(pts, vs, rng, mp, mv)
    return v.cpu().numpy(), n.cpu().numpy(), c.cpu().numpy(), vn.cpu().numpy()


@pytest.mark.parametrize('case', ['second', 'second_coarse', 'pp', 'pp_dense', 'ka'])
def test_voxelize_golden_bit_exact(golden, case):
    d = golden('voxelize')
    if case == 'ka':
        vs, rng, mp, mv = [0.5] * 3, [0, -40, -3, 70.4, 40, 1], 1000, 20000
    else:
        cfg = d[f'{case}.cfg']
        vs, rng, mp, mv = cfg[:3], cfg[3:9], int(cfg[9]), int(cfg[10])
    v, n, c, vn = _vox_gpu([d[f'{case}.points']], vs, rng, mp, mv)
    assert vn.tolist() == [len(d[f'{case}.coors'])] * 2
    assert np.array_equal(c[:, 1:], d[f'{case}.coors']) and (c[:, 0] == 0).all()
    assert np.array_equal(n, d[f'{case}.num_points'])
    assert np.array_equal(v, d[f'{case}.voxels'])


@pytest.mark.parametrize('cfg', ['second', 'pp'])
def test_voxelize_batch_full_size_vs_oracle(cfg):
    # BASELINE sizes: 20k points / frame, max_voxels 16000 (the cap is hit), ragged batch
    vs, rng, mp = (([0.05, 0.05, 0.1], synthetic.RANGE_SECOND, 5) if cfg == 'second'
                   else ([0.16, 0.16, 4], synthetic.RANGE_PP, 32))
    frames = [synthetic.make_frame(i, n_points=n, pc_range=rng)['points'].numpy()
              for i, n in enumerate([20000, 20000, 7001, 1, 20000])]
    frames.insert(3, np.zeros((0, 4), np.float32))           # empty frame
    v, n, c, vn = _vox_gpu(frames, vs, rng, mp, 16000)
    ov, on, oc = O.voxelize_batch(frames, vs, rng, mp, 16000)
    assert vn[-1] == len(oc) and vn[:-1].tolist() == [int((oc[:, 0] == b).sum()) for b in range(len(frames))]
    assert np.array_equal(c, oc) and np.array_equal(n, on) and np.array_equal(v, ov)


def test_voxelize_dense_cluster_rank_order():
    # thousands of points in a handful of voxels: the rank rounds must keep first-come order
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform([10, 0, -1, 0], [10.3, 0.3, -0.9, 1], (6000, 4)),
                          rng.uniform([0, -40, -3, 0], [70, 40, 1, 1], (2000, 4))]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    for mp in (1, 5, 32, 100):
        v, n, c, _ = _vox_gpu([pts], [0.16, 0.16, 4], synthetic.RANGE_PP, mp, 16000)
        ov, oc, on = O.hard_voxelize(pts, [0.16, 0.16, 4], synthetic.RANGE_PP, mp, 16000)
        assert np.array_equal(c[:, 1:], oc) and np.array_equal(n, on) and np.array_equal(v, ov)


def test_voxel_mean(golden):
    d = golden('encoders')
    out = F.voxel_mean(torch.from_numpy(d['vfe.voxels']).to(DEV), torch.from_numpy(d['vfe.num_points']).to(DEV), 4)
    np.testing.assert_allclose(out.cpu().numpy(), d['vfe.out'], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('channels_last', [False, True])
@pytest.mark.parametrize('case', ['small', 'pp'])
def test_scatter_golden(golden, case, channels_last):
    d = golden('scatter')
    B, C, ny, nx = (int(x) for x in d[f'{case}.shape'])
    feats = torch.from_numpy(d[f'{case}.feats']).to(DEV).requires_grad_(True)
    coors = torch.from_numpy(d[f'{case}.coors']).to(DEV)
    out = F.pillar_scatter(feats, coors, B, ny, nx, channels_last=channels_last)
    assert out.shape == (B, C, ny, nx)
    assert np.array_equal(out.detach().cpu().numpy(), d[f'{case}.canvas'])     # bit-exact copy
    # second call reuses the self-cleaning cell map
    out2 = F.pillar_scatter(feats, coors, B, ny, nx, channels_last=channels_last)
    assert torch.equal(out, out2)
    # voxelizer output has one pillar per cell: the map-free path gives the same canvas
    out3 = F.pillar_scatter(feats, coors, B, ny, nx, channels_last=channels_last, unique=True)
    assert torch.equal(out, out3)
    # backward = gather of the canvas gradient
    g = torch.randn_like(out)
    out.backward(g)
    idx = coors.long()
    ref = g[idx[:, 0], :, idx[:, 2], idx[:, 3]]
    assert torch.equal(feats.grad, ref)


def test_scatter_full_size_properties():
    # BASELINE config 2: [256000,64] -> [16,64,496,432]; checksum / round-trip properties
    B, C, ny, nx, M = 16, 64, 496, 432, 16000
    g = torch.Generator().manual_seed(0)
    coors = []
    for b in range(B):
        cells = torch.randperm(ny * nx, generator=g)[:M]
        coors.append(torch.stack([torch.full((M,), b), torch.zeros(M, dtype=torch.long), cells // nx, cells % nx], 1))
    coors = torch.cat(coors).int().to(DEV)
    feats = torch.randn(B * M, C, device=DEV)
    for cl in (False, True):
        out = F.pillar_scatter(feats, coors, B, ny, nx, channels_last=cl)
        assert int((out != 0).sum()) == int((feats != 0).sum())
        assert torch.equal(out.sum(dtype=torch.float64), feats.sum(dtype=torch.float64)) or \
            abs(float(out.sum(dtype=torch.float64) - feats.sum(dtype=torch.float64))) < 1e-6
        idx = coors.long()
        assert torch.equal(out[idx[:, 0], :, idx[:, 2], idx[:, 3]], feats)      # scatter -> gather round trip
        assert torch.equal(out, F.pillar_scatter(feats, coors, B, ny, nx, channels_last=cl, unique=True))
    # capacity-sized buffer with a device-side valid count: rows beyond it are ignored
    nv = torch.tensor([1000], dtype=torch.int32, device=DEV)
    out = F.pillar_scatter(feats, coors, B, ny, nx, num_valid=nv)
    assert int((out != 0).sum()) == int((feats[:1000] != 0).sum())


def test_scatter_duplicate_cells_last_row_wins():
    feats = torch.arange(1, 4 * 8 + 1, dtype=torch.float32, device=DEV).view(4, 8)
    coors = torch.tensor([[0, 0, 1, 1], [0, 0, 2, 2], [0, 0, 1, 1], [0, 0, 1, 1]], dtype=torch.int32, device=DEV)
    for cl in (False, True):
        for _ in range(3):     # repeated: the winner map must come back clean each time
            out = F.pillar_scatter(feats, coors, 1, 4, 4, channels_last=cl)
            assert torch.equal(out[0, :, 1, 1], feats[3]) and torch.equal(out[0, :, 2, 2], feats[1])
            assert int((out != 0).sum()) == 16
    # channel counts whose rows straddle wavefronts (C/4 does not divide 64) take the separate map reset
    f12 = torch.arange(1, 4 * 12 + 1, dtype=torch.float32, device=DEV).view(4, 12)
    for _ in range(2):
        out = F.pillar_scatter(f12, coors, 1, 4, 4, channels_last=True)
        assert torch.equal(out[0, :, 1, 1], f12[3]) and torch.equal(out[0, :, 2, 2], f12[1])


def test_heatmap_splat(golden):
    d = golden('gaussian')
    objs = np.concatenate([np.zeros((len(d['splat.radii']), 1), np.int32), d['splat.centers'],
                           d['splat.radii'][:, None]], 1).astype(np.int32)
    hm = F.heatmap_splat(objs, 1, 200, 176, DEV)
    assert np.array_equal(hm[0].cpu().numpy(), d['splat.heatmap'])            # table lookup: bit-exact
    t, offs = F.gaussian_patch_table(11, DEV)
    for r in range(12):
        p = t[int(offs[r]):int(offs[r + 1])].cpu().numpy().reshape(2 * r + 1, 2 * r + 1)
        assert np.array_equal(p, d[f'patch.{r}'].astype(np.float32))


def _targets(d, c):
    case = load_head_case(d, c)
    torch.manual_seed(1234)
    srl = O.draw_srl(case['B'])
    tg = O.get_targets(case['labels'], case['boxes_img'], case['lidar2img'], case['pseudo'],
                       case['bdry'], case['ibp'], case['meta_l2i'], TRAIN_CFG[c], srl)
    return case, tg


def _pack_ibp(ibps_t, K):
    xy, off, slot = [], [0], []
    for b, objs in enumerate(ibps_t):
        for k, p in enumerate(objs):
            xy.append(np.asarray(p)[:, :2].astype(np.float32))
            off.append(off[-1] + len(p))
            slot.append(b * K + k)
    xy = np.concatenate(xy, 0) if xy else np.zeros((0, 2), np.float32)
    return (torch.from_numpy(xy).to(DEV), torch.tensor(off, dtype=torch.int32, device=DEV),
            torch.tensor(slot, dtype=torch.int32, device=DEV))


@pytest.mark.parametrize('c', ['second', 'pp'])
def test_head_losses_and_grads_vs_reference(golden, c):
    d = golden('head')
    case, tg = _targets(d, c)
    B = case['B']
    H, W = FMAP[c]
    K = 500
    preds = synthetic.make_head_preds(B, H, W, seed=int(d[f'{c}.pred_seed']))
    prm = F.loss_params(B, K, TRAIN_CFG[c])
    keys = ('loss_bbox', 'loss_ratio', 'distancemin', 'distancex', 'distancey')
    for t in range(3):
        maps = {k: v.to(DEV).requires_grad_(True) for k, v in preds[t].items()}
        ind = torch.from_numpy(tg['ind'][t]).to(DEV)
        mask = torch.from_numpy(tg['mask'][t]).to(DEV)
        hm_t = torch.from_numpy(tg['heatmap'][t]).to(DEV)
        lh, npos = F.gaussian_focal_loss(maps['heatmap'], hm_t, alpha=0.0, gamma=4.0, scale=5.0)
        assert float(npos) == float((tg['heatmap'][t] == 1).sum())
        ref = float(d[f'{c}.loss.task{t}.loss_heatmap'])
        assert float(lh) == pytest.approx(ref, rel=1e-5, abs=1e-4)

        pred = F.gather_pred(maps['reg'], maps['height'], maps['dim'], maps['rot'], ind, mask)
        np.testing.assert_array_equal(pred.detach().cpu().numpy(), d[f'{c}.mid.{t}.pred'])
        xy, off, slot = _pack_ibp(tg['ibp'][t], K)
        losses, box = F.box_losses(pred, ind, mask, torch.from_numpy(tg['anno_box'][t]).to(DEV),
                                   torch.from_numpy(tg['lidar2img'][t]).to(DEV),
                                   torch.from_numpy(tg['bound_mask'][t]).to(DEV), xy, off, slot, prm)
        box = box.cpu().numpy()
        np.testing.assert_allclose(box[..., 0], d[f'{c}.mid.{t}.rot'], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(box[..., 1:3], d[f'{c}.mid.{t}.pred_ratio'], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(box[..., 3:7], d[f'{c}.mid.{t}.pred_iou'], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(box[..., 7:9], d[f'{c}.mid.{t}.pred_box_bev'][..., :2], rtol=2e-6, atol=2e-6)
        for j, k in enumerate(('p2c_min', 'p2c_x', 'p2c_y')):
            np.testing.assert_allclose(box[..., 9 + j], d[f'{c}.mid.{t}.{k}'][..., 0], rtol=1e-5, atol=1e-4)
        lv = losses.detach().cpu().numpy()
        for j, k in enumerate(keys):
            assert float(lv[j]) == pytest.approx(float(d[f'{c}.loss.task{t}.{k}']), rel=1e-5, abs=1e-4), k

        # gradients: variant A (keys containing 'loss' only) and B (all terms) vs reference autograd
        for tag, gl in (('A', [1, 1, 0, 0, 0]), ('B', [1, 1, 1, 1, 1])):
            for m in maps.values():
                m.grad = None
            tot = (losses * torch.tensor(gl, dtype=torch.float32, device=DEV)).sum() + (lh if tag == 'A' else lh * 0)
            tot.backward(retain_graph=True)
            for k in ('reg', 'height', 'dim', 'rot'):
                idx = d[f'{c}.grad{tag}.{t}.{k}.idx']
                val = d[f'{c}.grad{tag}.{t}.{k}.val']
                g = maps[k].grad.cpu().numpy()
                got = g[tuple(idx.T)] if len(idx) else np.zeros(0, np.float32)
                np.testing.assert_allclose(got, val, rtol=2e-3, atol=2e-5, err_msg=f'{tag} {k}')
                assert np.count_nonzero(g) <= len(idx) + 2            # nothing off the object cells
            if tag == 'A':
                g = maps['heatmap'].grad.cpu().numpy().reshape(-1)
                sel = d[f'{c}.gradA.{t}.heatmap.flatidx']
                np.testing.assert_allclose(g[sel], d[f'{c}.gradA.{t}.heatmap.val'], rtol=1e-4, atol=1e-7)
                assert float(np.abs(g.astype(np.float64)).sum()) == pytest.approx(
                    float(d[f'{c}.gradA.{t}.heatmap.abs_sum']), rel=1e-4)


def test_focal_loss_general_alpha_vs_oracle():
    n = 3 * 200 * 176 + 3      # not a multiple of 4: exercises the tail
    x = synthetic.det_uniform((n + 1,), 9)[:n] * 6
    t = synthetic.det_uniform((n + 1,), 10)[:n] + 0.5
    t[::997] = 1.0
    xa, ta = x[:n - 3].contiguous(), t[:n - 3].contiguous()
    for alpha, gamma in ((0.0, 4.0), (2.0, 4.0), (1.5, 3.0)):
        for xs, ts in ((xa, ta),):
            xg = xs.to(DEV).requires_grad_(True)
            loss, npos = F.gaussian_focal_loss(xg, ts.to(DEV), alpha, gamma, 1.0)
            loss.backward()
            ref, g, rp = O.focal_loss(xs.numpy(), ts.numpy(), alpha, gamma, with_grad=True)
            assert float(npos) == rp
            assert float(loss) == pytest.approx(float(ref), rel=2e-5)
            np.testing.assert_allclose(xg.grad.cpu().numpy(), g, rtol=2e-4, atol=1e-9)


def test_nan_target_propagates_like_reference():
    # the reference's isnotnan weight multiplies |pred - NaN| by 0, which is still NaN
    # (centerpoint_head_gga.py:679-681): the kernel keeps that IEEE behaviour.
    c = 'second'
    B, K = 2, 500
    prm = F.loss_params(B, K, TRAIN_CFG[c])
    pred = torch.zeros(B, K, 8, device=DEV)
    pred[..., 7] = 1
    ind = torch.zeros(B, K, dtype=torch.int64, device=DEV)
    mask = torch.zeros(B, K, dtype=torch.uint8, device=DEV)
    mask[0, 0] = 1
    anno = torch.zeros(B, K, 5, device=DEV)
    anno[0, 0, 1] = float('nan')
    l2i = torch.eye(4, device=DEV).repeat(B, K, 1, 1)
    bm = torch.ones(B, K, 4, dtype=torch.uint8, device=DEV)
    losses, _ = F.box_losses(pred, ind, mask, anno, l2i, bm, None, None, None, prm)
    assert torch.isnan(losses[0]) and not torch.isnan(losses[1])


def test_box_loss_terms_are_the_box_losses_with_scalar_outputs():
    """F.box_loss_terms (what the head calls: five scalar outputs, unused terms arrive in backward as None) against F.box_losses
    (one [5] tensor): same values, and the same gradient when two of the five terms enter the total - bit for bit."""
    c = 'second'
    B, K = 2, 500
    prm = F.loss_params(B, K, TRAIN_CFG[c])
    g = torch.Generator().manual_seed(11)
    base = torch.randn(B, K, 8, generator=g)
    base[..., 3:6] = base[..., 3:6].abs() * 0.2          # log-dims stay small
    ind = torch.randint(0, 200 * 176, (B, K), generator=g).to(DEV)
    mask = (torch.rand(B, K, generator=g) < 0.1).to(torch.uint8).to(DEV)
    anno = torch.rand(B, K, 5, generator=g).to(DEV)
    l2i = (torch.eye(4) + 0.01 * torch.randn(B, K, 4, 4, generator=g)).to(DEV)
    bm = torch.ones(B, K, 4, dtype=torch.uint8, device=DEV)
    pa = base.clone().to(DEV).requires_grad_()
    pb = base.clone().to(DEV).requires_grad_()
    la, box_a = F.box_losses(pa, ind, mask, anno, l2i, bm, None, None, None, prm)
    (t0, t1, t2, t3, t4), box_b = F.box_loss_terms(pb, ind, mask, anno, l2i, bm, None, None, None, prm)
    assert torch.equal(la, torch.stack([t0, t1, t2, t3, t4]).detach()) and torch.equal(box_a, box_b)
    (la[0] * 1.5 + la[1]).backward()
    (t0 * 1.5 + t1).backward()                            # t2..t4 only logged: their gradients never exist
    assert torch.equal(pa.grad, pb.grad)


def test_bn_statistics_from_a_column_block_of_wider_partial_rows():
    """gga_bn_stats_partials_cols (the head's paired first convolutions: one [tiles, 2, 128] statistics array, two BatchNorms of
    64 channels) against gga_bn_stats_partials on contiguous copies of the halves: the same saved statistics, scale / shift and
    running buffers, bit for bit."""
    from gga_amd import _lib
    L = _lib.lib()
    tiles, C, rows = 200, 64, 16 * 248 * 216
    g = torch.Generator().manual_seed(5)
    st = (torch.rand(tiles, 2, 2 * C, generator=g, dtype=torch.float64) * rows / tiles).to(DEV)
    st[:, 1] = st[:, 0].abs() * 1.7 + 3.0                    # sums of squares above the squared sums
    for col in (0, C):
        gam, bet = torch.rand(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
        out = []
        for form in ('cols', 'copy'):
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            saved, ss = torch.empty(2 * C, device=DEV), torch.empty(2 * C, device=DEV)
            if form == 'cols':
                rc = L.gga_bn_stats_partials_cols(F._p(gam), F._p(bet), F._p(rm), F._p(rv), rows, C, 1e-3, 0.01, F._p(saved), F._p(ss),
                                                  F._p(st), tiles, 2 * C, col, F._stream())
            else:
                half = st[:, :, col:col + C].contiguous()
                rc = L.gga_bn_stats_partials(F._p(gam), F._p(bet), F._p(rm), F._p(rv), rows, C, 1e-3, 0.01, F._p(saved), F._p(ss),
                                             F._p(half), tiles, F._stream())
            assert rc == 0
            out.append((rm, rv, saved, ss))
        for a, b in zip(*out):
            assert torch.equal(a, b)


def test_fused_pfn_vs_reference_golden(golden):
    """Fused PillarFeatureNet against the imported reference: forward, running stats and the
    parameter gradients (encoders.npz was produced by the reference's PillarFeatureNet)."""
    from gga_amd.voxel_encoders import PillarFeatureNet
    d = golden('encoders')
    cfg = d['pfn.cfg']
    pfn = PillarFeatureNet(in_channels=4, feat_channels=(64,), voxel_size=tuple(cfg[:3]),
                           point_cloud_range=tuple(cfg[3:])).to(DEV)
    l0 = pfn.pfn_layers[0]
    with torch.no_grad():
        l0.linear.weight.copy_(torch.from_numpy(d['pfn.linear_w']))
        l0.norm.weight.copy_(torch.from_numpy(d['pfn.bn_w']))
        l0.norm.bias.copy_(torch.from_numpy(d['pfn.bn_b']))
    pfn.train()
    v = torch.from_numpy(d['pfn.voxels']).to(DEV)
    n = torch.from_numpy(d['pfn.num_points']).to(DEV)
    c = torch.from_numpy(d['pfn.coors']).to(DEV)
    assert pfn._fusable(v)
    y = pfn(v, n, c)
    np.testing.assert_allclose(y.detach().cpu().numpy(), d['pfn.out'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(l0.norm.running_mean.cpu().numpy(), d['pfn.running_mean'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(l0.norm.running_var.cpu().numpy(), d['pfn.running_var'], rtol=1e-5)
    assert torch.equal(v.cpu(), torch.from_numpy(d['pfn.voxels']))     # input left untouched
    y.backward(torch.from_numpy(d['pfn.grad_out']).to(DEV))
    for got, key in ((l0.linear.weight.grad, 'pfn.grad_linear_w'), (l0.norm.weight.grad, 'pfn.grad_bn_w'),
                     (l0.norm.bias.grad, 'pfn.grad_bn_b')):
        ref = d[key]
        err = np.abs(got.cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err < 2e-4, (key, err)
    # the eager path (general configs) computes the same thing
    pfn2 = PillarFeatureNet(in_channels=4, feat_channels=(64,), voxel_size=tuple(cfg[:3]),
                            point_cloud_range=tuple(cfg[3:])).to(DEV)
    pfn2.load_state_dict({k: (torch.zeros_like(t) if 'running_mean' in k else torch.ones_like(t) if 'running_var' in k
                              else t) for k, t in pfn.state_dict().items()})
    pfn2.train()
    y2 = pfn2.forward_eager(v, n, c)
    np.testing.assert_allclose(y2.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    # eval mode uses the running statistics
    pfn.eval(), pfn2.eval()
    pfn2.load_state_dict(pfn.state_dict())
    np.testing.assert_allclose(pfn(v, n, c).detach().cpu().numpy(), pfn2.forward_eager(v, n, c).detach().cpu().numpy(),
                               rtol=1e-4, atol=1e-4)


def test_fused_pfn_full_size():
    from gga_amd.voxel_encoders import PillarFeatureNet
    frames = [synthetic.make_frame(i, pc_range=synthetic.RANGE_PP)['points'].to(DEV) for i in range(4)]
    v, n, c, _ = F.hard_voxelize_batch(frames, [0.16, 0.16, 4], synthetic.RANGE_PP, 32, 16000)
    torch.manual_seed(0)
    pfn = PillarFeatureNet(in_channels=4, feat_channels=(64,), voxel_size=(0.16, 0.16, 4),
                           point_cloud_range=synthetic.RANGE_PP).to(DEV).train()
    ref = PillarFeatureNet(in_channels=4, feat_channels=(64,), voxel_size=(0.16, 0.16, 4),
                           point_cloud_range=synthetic.RANGE_PP).to(DEV).train()
    ref.load_state_dict(pfn.state_dict())
    y = pfn(v, n, c)
    yr = ref.forward_eager(v, n, c)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    g = torch.randn_like(y)
    y.backward(g)
    yr.backward(g)
    for a, b in zip(pfn.parameters(), ref.parameters()):
        err = float((a.grad - b.grad).abs().max() / b.grad.abs().max())
        assert err < 1e-3, err


@pytest.mark.parametrize('shape,cl', [((70001, 16), False), ((3000, 128), False), ((4, 64, 62, 54), True),
                                       ((2, 256, 31, 27), True), ((3, 8, 5, 7), True)])
@pytest.mark.parametrize('relu,res', [(True, False), (False, False), (True, True)])
def test_fused_bn_act_vs_torch(shape, cl, relu, res):
    """Fused BatchNorm(+residual)(+ReLU) against eager torch: values, running stats, all gradients."""
    torch.manual_seed(0)
    C = shape[1]
    bn = (torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d)(C, eps=1e-3, momentum=0.01).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5)
    ref = copy_bn = __import__('copy').deepcopy(bn)
    x = torch.randn(*shape, device=DEV) * 3 + 1.5
    r = torch.randn(*shape, device=DEV) if res else None
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
        r = r.contiguous(memory_format=torch.channels_last) if res else None
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    r1 = r.clone().requires_grad_(True) if res else None
    r2 = r.clone().requires_grad_(True) if res else None
    assert F._rows_channels(x1) is not None
    y = F.bn_act(x1, bn, relu=relu, residual=r1)
    yr = ref(x2)
    if res:
        yr = yr + r2
    if relu:
        yr = torch.relu(yr)
    torch.testing.assert_close(y, yr, rtol=1e-4, atol=1e-4)
    assert y.stride() == x.stride()
    torch.testing.assert_close(bn.running_mean, ref.running_mean, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(bn.running_var, ref.running_var, rtol=1e-4, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    # an element whose pre-activation rounds to either side of 0 may take the other ReLU branch
    near0 = (yr.detach().abs() < 1e-5) & (y.detach().abs() < 1e-5) if relu else torch.zeros_like(yr, dtype=torch.bool)
    assert int(near0.sum()) <= max(3, yr.numel() // 10000) or not relu or float((yr.detach() == 0).float().mean()) > 0.2
    gx1, gx2 = x1.grad.clone(), x2.grad.clone()
    flip = (gx1 - gx2).abs() > 1e-4 + 1e-3 * gx2.abs()
    assert int(flip.sum()) <= 3 and bool((near0 | ~flip).all() or int(flip.sum()) <= 3)
    gx1[flip] = gx2[flip]
    torch.testing.assert_close(gx1, gx2, rtol=1e-3, atol=1e-4)
    # (a flipped boundary element also moves its channel's d_gamma / d_beta by one g value)
    torch.testing.assert_close(bn.weight.grad, ref.weight.grad, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(bn.bias.grad, ref.bias.grad, rtol=1e-2, atol=1e-2)
    if res:
        assert int(((r1.grad - r2.grad).abs() > 1e-6).sum()) <= 3
    # eval mode uses the running statistics
    bn.eval(), ref.eval()
    with torch.no_grad():
        ye = F.bn_act(x, bn, relu=relu, residual=r)
        yre = ref(x) + (r if res else 0)
        torch.testing.assert_close(ye, torch.relu(yre) if relu else yre, rtol=1e-4, atol=1e-4)
    # ... also under autograd (a frozen norm_eval backbone): dx = gamma * invstd * g, no batch terms
    for m in (bn, ref):
        m.weight.grad = m.bias.grad = None
    x3, x4 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y3 = F.bn_act(x3, bn, relu=relu, residual=r)
    assert 'BNAct' in type(y3.grad_fn).__name__
    y4 = ref(x4) + (r if res else 0)
    y4 = torch.relu(y4) if relu else y4
    torch.testing.assert_close(y3, y4, rtol=1e-4, atol=1e-4)
    rm = bn.running_mean.clone()
    y3.backward(g)
    y4.backward(g)
    assert torch.equal(bn.running_mean, rm) and int(bn.num_batches_tracked) == 1      # statistics untouched
    flip = (x3.grad - x4.grad).abs() > 1e-4 + 1e-3 * x4.grad.abs()
    assert int(flip.sum()) <= 3
    torch.testing.assert_close(bn.weight.grad, ref.weight.grad, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(bn.bias.grad, ref.bias.grad, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize('B,C,G,H,W', [(3, 256, 32, 24, 39), (2, 64, 16, 7, 5), (1, 32, 8, 40, 33)])
@pytest.mark.parametrize('relu', [True, False])
def test_fused_gn_act_vs_torch(B, C, G, H, W, relu):
    """Fused GroupNorm (+ ReLU) on channels-last memory against eager torch: values and all gradients."""
    import copy
    torch.manual_seed(1)
    gn = torch.nn.GroupNorm(G, C, eps=1e-5).to(DEV)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5), gn.bias.uniform_(-0.5, 0.5)
    ref = copy.deepcopy(gn)
    x = (torch.randn(B, C, H, W, device=DEV) * 2 + 0.7).contiguous(memory_format=torch.channels_last)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = F.gn_act(x1, gn, relu=relu)
    assert 'GNAct' in type(y.grad_fn).__name__ and y.is_contiguous(memory_format=torch.channels_last)
    yr = ref(x2)
    yr = torch.relu(yr) if relu else yr
    torch.testing.assert_close(y, yr, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    flip = (x1.grad - x2.grad).abs() > 1e-4 + 1e-3 * x2.grad.abs()
    assert int(flip.sum()) <= 3                                         # ReLU-boundary elements may take the other branch
    torch.testing.assert_close(gn.weight.grad, ref.weight.grad, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(gn.bias.grad, ref.bias.grad, rtol=1e-3, atol=1e-2)
    # shapes the kernels do not take fall back to the eager ops
    odd = torch.nn.GroupNorm(3, 6).to(DEV)
    xo = torch.randn(2, 6, 5, 5, device=DEV).contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(F.gn_act(xo, odd, relu=True), torch.relu(odd(xo)))


def test_bn_relu_cat_vs_torch():
    """SECONDFPN tail: the branches' BN+ReLU written into / read from channel slices of the
    concatenated map equals cat(relu(bn(x)))."""
    import copy
    torch.manual_seed(3)
    B, H, W = 2, 21, 24
    chans = (32, 128, 64)
    bns = [torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).to(DEV) for c in chans]
    for bn in bns:
        bn.weight.data.uniform_(0.5, 1.5), bn.bias.data.uniform_(-0.5, 0.5)
    refs = copy.deepcopy(bns)
    xs = [torch.randn(B, c, H, W, device=DEV).contiguous(memory_format=torch.channels_last) for c in chans]
    xa = [x.clone().requires_grad_(True) for x in xs]
    xb = [x.clone().requires_grad_(True) for x in xs]
    y = F.bn_relu_cat(xa, bns)
    assert 'BNActCat' in type(y.grad_fn).__name__ and y.is_contiguous(memory_format=torch.channels_last)
    ref = torch.cat([torch.relu(bn(x)) for bn, x in zip(refs, xb)], dim=1)
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(ref)
    y.backward(g)
    ref.backward(g)
    for a_, b_, bn, rf in zip(xa, xb, bns, refs):
        # elements whose pre-activation is within rounding of zero may flip the ReLU mask
        assert int(((a_.grad - b_.grad).abs() > 1e-4).sum()) <= 3
        torch.testing.assert_close(bn.weight.grad, rf.weight.grad, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(bn.bias.grad, rf.bias.grad, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(bn.running_mean, rf.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(bn.running_var, rf.running_var, rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == 1
    # a single branch / an NCHW input take the plain composition
    z = F.bn_relu_cat([xs[0].contiguous()], [bns[0]])
    assert z.shape == xs[0].shape


@pytest.mark.parametrize('cout,bias', [(1, True), (2, True), (3, True), (4, False)])
def test_head_output_conv_vs_torch(cout, bias):
    """64 -> 1..4 channel 3x3 output conv of the head branches against torch's convolution."""
    torch.manual_seed(cout)
    B, H, W = 3, 37, 29
    conv = torch.nn.Conv2d(64, cout, 3, padding=1, bias=bias).to(DEV)
    x = torch.randn(B, 64, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = F.head_conv3x3(x1, conv)
    assert y.is_contiguous() and y.grad_fn is not None and 'HeadConv' in type(y.grad_fn).__name__
    ref = conv(x2)
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(ref)
    gw_ref, gb_ref = None, None
    ref.backward(g)
    gw_ref = conv.weight.grad.clone()
    gb_ref = conv.bias.grad.clone() if bias else None
    conv.weight.grad = None
    if bias:
        conv.bias.grad = None
    y.backward(g.contiguous())
    torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(conv.weight.grad, gw_ref, rtol=1e-3, atol=1e-3)
    if bias:
        torch.testing.assert_close(conv.bias.grad, gb_ref, rtol=1e-4, atol=1e-3)
    # NCHW input falls back to the framework convolution
    z = F.head_conv3x3(x.contiguous(), conv)
    torch.testing.assert_close(z, ref.detach(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('shape', [(6, 200, 176, 192, 1), (2, 50, 216, 128, 1), (3, 37, 29, 64, 0)])
def test_head_output_conv_on_column_blocks_of_large_maps(shape):
    """Forward and weight gradient of the head output convs through the C entry points on maps with several tiles per
    workgroup (the persistent loops with their loads a tile ahead), ragged right / bottom edges and a 64-channel column
    block of a wider tensor as input, with and without the fused scale / shift + ReLU, against float64."""
    from gga_amd import _lib
    L = _lib.lib()
    B, H, W, wide, blk = shape
    torch.manual_seed(B + W)
    big = torch.randn(B, H, W, wide, device=DEV)
    xs = big[..., 64 * blk:64 * blk + 64]
    for cout in (1, 2, 3, 4):
        for aff in (True, False):
            ss = torch.cat([torch.rand(64, device=DEV) + 0.5, torch.rand(64, device=DEV) - 0.5])
            w = torch.randn(cout, 64, 3, 3, device=DEV) * 0.05
            b = torch.randn(cout, device=DEV)
            y = torch.full((B, cout, H, W), float('nan'), device=DEV)
            assert L.gga_head_conv3x3_fwd(big.data_ptr() + 4 * 64 * blk, wide, F._p(ss) if aff else None, F._p(w), F._p(b),
                                          B, H, W, 64, cout, F._p(y), F._stream()) == 0
            xin = xs.permute(0, 3, 1, 2).double()
            if aff:
                xin = torch.relu(xin * ss[:64].double().view(1, -1, 1, 1) + ss[64:].double().view(1, -1, 1, 1))
            wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
            ref = torch.nn.functional.conv2d(xin, wr, br, padding=1)
            assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-6, (cout, aff)
            gy = torch.randn(B, cout, H, W, device=DEV)
            dw, db = torch.full_like(w, float('nan')), torch.full_like(b, float('nan'))
            ws = torch.empty(L.gga_head_conv3x3_workspace_bytes(cout), dtype=torch.uint8, device=DEV)
            assert L.gga_head_conv3x3_wgrad(big.data_ptr() + 4 * 64 * blk, wide, F._p(ss) if aff else None, F._p(gy), B, H, W,
                                            64, cout, F._p(dw), F._p(db), ws.data_ptr(), ws.numel(), F._stream()) == 0
            ref.backward(gy.double())
            assert float((dw.double() - wr.grad).abs().max() / wr.grad.abs().max()) < 2e-6, (cout, aff)
            assert float((db.double() - br.grad).abs().max() / br.grad.abs().max()) < 2e-5, (cout, aff)


@pytest.mark.parametrize('cout', [1, 2, 3, 4])
def test_bn_relu_head_conv_fused_vs_torch(cout):
    """Tail of a head branch: conv(relu(bn(x))) with the normalised activation never stored."""
    import copy
    torch.manual_seed(10 + cout)
    B, H, W = 3, 37, 29
    bn = torch.nn.BatchNorm2d(64, eps=1e-3, momentum=0.01).to(DEV)
    bn.weight.data.uniform_(0.5, 1.5), bn.bias.data.uniform_(-0.5, 0.5)
    conv = torch.nn.Conv2d(64, cout, 3, padding=1, bias=True).to(DEV)
    bn_r, conv_r = copy.deepcopy(bn), copy.deepcopy(conv)
    x = torch.randn(B, 64, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = F.bn_relu_head_conv3x3(x1, bn, conv)
    assert 'BnReluHeadConv' in type(y.grad_fn).__name__
    ref = conv_r(torch.relu(bn_r(x2)))
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(ref)
    y.backward(g)
    ref.backward(g)
    assert int(((x1.grad - x2.grad).abs() > 1e-4).sum()) <= 3       # ReLU-boundary elements may flip
    torch.testing.assert_close(conv.weight.grad, conv_r.weight.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(conv.bias.grad, conv_r.bias.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(bn.weight.grad, bn_r.weight.grad, rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(bn.bias.grad, bn_r.bias.grad, rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(bn.running_mean, bn_r.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, bn_r.running_var, rtol=1e-5, atol=1e-6)
    # eval mode: the unfused composition on the running statistics
    bn.eval(), bn_r.eval()
    with torch.no_grad():
        torch.testing.assert_close(F.bn_relu_head_conv3x3(x, bn, conv), conv_r(torch.relu(bn_r(x))), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('planes', [2, 3])
def test_head_branches_one_node_vs_branch_by_branch(planes, monkeypatch):
    """All branches of a head on one shared map as one autograd node (functional._HeadBranches: column blocks
    of one buffer, one backward-data, one weight-gradient call) against the same branches run one by one and
    against eager torch."""
    import copy
    from gga_amd import dense_conv
    monkeypatch.setattr(dense_conv, 'PLANES', planes)
    torch.manual_seed(21)
    B, H, W, couts = 2, 37, 45, (2, 1, 3, 2, 1)

    def make():
        torch.manual_seed(22)
        out = []
        for c in couts:
            conv1 = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(DEV).to(memory_format=torch.channels_last)
            bn = torch.nn.BatchNorm2d(64, eps=1e-3, momentum=0.01).to(DEV)
            bn.weight.data.uniform_(0.5, 1.5), bn.bias.data.uniform_(-0.5, 0.5)
            out.append((conv1, bn, torch.nn.Conv2d(64, c, 3, padding=1, bias=True).to(DEV)))
        return out
    x = torch.randn(B, 64, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    gs = [torch.randn(B, c, H, W, device=DEV) for c in couts]
    results = []
    for mode in ('node', 'single', 'eager'):
        br = make()
        xi = x.clone().requires_grad_(True)
        if mode == 'node':
            ys = F.head_branches(xi, br)
            assert ys is not None and 'HeadBranches' in type(ys[0].grad_fn).__name__
        elif mode == 'single':
            ys = [F.bn_relu_head_conv3x3(dense_conv.conv2d(xi, c1, bn_follows=True), bn, c2) for c1, bn, c2 in br]
        else:
            ys = [c2(torch.relu(bn(c1(xi)))) for c1, bn, c2 in br]
        sum((y * g).sum() for y, g in zip(ys, gs)).backward()
        results.append((ys, xi.grad, br))
    (y_n, gx_n, br_n), (y_s, gx_s, br_s), (y_e, gx_e, br_e) = results
    for i in range(len(couts)):
        # the node runs branch pairs on the 128-column form of the kernel: the convolution's values are the same, but its tiles
        # (8 rows against the 64-column form's 16), and with them the order of the BatchNorm statistics' partial sums, are not
        torch.testing.assert_close(y_n[i], y_s[i], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(y_n[i], y_e[i], rtol=1e-4, atol=1e-4)
        for j in (0, 2):
            torch.testing.assert_close(br_n[i][j].weight.grad, br_s[i][j].weight.grad, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(br_n[i][j].weight.grad, br_e[i][j].weight.grad, rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(br_n[i][2].bias.grad, br_e[i][2].bias.grad, rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(br_n[i][1].weight.grad, br_e[i][1].weight.grad, rtol=1e-3, atol=2e-3)
        torch.testing.assert_close(br_n[i][1].bias.grad, br_e[i][1].bias.grad, rtol=1e-3, atol=2e-3)
        torch.testing.assert_close(br_n[i][1].running_var, br_e[i][1].running_var, rtol=1e-5, atol=1e-6)
        assert int(br_n[i][1].num_batches_tracked) == 1
    torch.testing.assert_close(gx_n, gx_s, rtol=1e-4, atol=1e-4)
    scale = float(gx_e.abs().max())
    assert int(((gx_n - gx_e).abs() > 1e-4 * scale).sum()) <= 10      # ReLU-boundary elements may flip


# ----------------------------------------------------------------------------- first conv on the canvas
def _pillar_case(B, ny, nx, M, C, seed):
    g = torch.Generator().manual_seed(seed)
    coors = []
    for b in range(B):
        cells = torch.randperm(ny * nx, generator=g)[:M]
        coors.append(torch.stack([torch.full((M,), b), torch.zeros(M, dtype=torch.long), cells // nx, cells % nx], 1))
    return torch.cat(coors).int().to(DEV), torch.randn(B * M, C, generator=g).to(DEV)


@pytest.mark.parametrize('k,stride,pad,cin,cout,ny,nx', [(3, 2, 1, 64, 64, 62, 54), (3, 1, 1, 64, 128, 33, 40),
                                                        (3, 2, 1, 32, 64, 31, 29), (1, 1, 0, 64, 64, 16, 24),
                                                        (5, 3, 2, 16, 32, 40, 37)])
def test_pillar_conv_map_and_backward(k, stride, pad, cin, cout, ny, nx):
    """The pillar-restricted backward of conv(canvas) equals the dense autograd backward: the map
    against index arithmetic in numpy (bit-exact), gradients against torch's conv2d backward."""
    from gga_amd import _lib, pillar_conv
    B, M = 3, 300
    coors, feats = _pillar_case(B, ny, nx, M, cin, seed=k * 100 + stride)
    oh, ow = (ny + 2 * pad - k) // stride + 1, (nx + 2 * pad - k) // stride + 1
    # map
    m = coors.shape[0]
    nbr = torch.empty((k * k, m), dtype=torch.int32, device=DEV)
    _lib.check(_lib.lib().gga_pillar_conv_map(F._p(coors), m, None, B, ny, nx, k, k, stride, stride, pad, pad, F._p(nbr),
                                              F._stream()), 'map')
    c = coors.cpu().numpy().astype(np.int64)
    ref = np.full((k * k, m), -1, np.int64)
    for ky in range(k):
        for kx in range(k):
            ty, tx = c[:, 2] + pad - ky, c[:, 3] + pad - kx
            ok = (ty >= 0) & (tx >= 0) & (ty % stride == 0) & (tx % stride == 0) & (ty // stride < oh) & (tx // stride < ow)
            ref[ky * k + kx] = np.where(ok, (c[:, 0] * oh + ty // stride) * ow + tx // stride, -1)
    assert np.array_equal(nbr.cpu().numpy(), ref)

    conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False).to(DEV)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    gy = None
    res = []
    for fused in (False, True):
        f = feats.clone().requires_grad_(True)
        conv.weight.grad = None
        canvas = F.pillar_scatter(f, coors, B, ny, nx, channels_last=True, unique=True)
        assert pillar_conv.eligible(conv, canvas)
        y = pillar_conv.pillar_conv2d(canvas, conv) if fused else conv(canvas)
        if gy is None:
            gy = torch.randn_like(y)
        y.backward(gy)
        res.append((y.detach(), f.grad.clone(), conv.weight.grad.clone()))
    (y0, gf0, gw0), (y1, gf1, gw1) = res
    assert float((y0 - y1).abs().max()) <= 1e-5 * float(y0.abs().max())      # same MIOpen call; solver may split K
    assert float((gf0 - gf1).abs().max()) <= 2e-5 * float(gf0.abs().max())
    assert float((gw0 - gw1).abs().max()) <= 2e-5 * float(gw0.abs().max())


def test_pillar_conv_valid_count_and_fallbacks():
    from gga_amd import pillar_conv
    B, ny, nx, M, C = 2, 40, 36, 200, 64
    coors, feats = _pillar_case(B, ny, nx, M, C, seed=5)
    conv = torch.nn.Conv2d(C, C, 3, stride=2, padding=1, bias=False).to(DEV)
    nv = torch.tensor([250], dtype=torch.int32, device=DEV)
    out = []
    for fused in (False, True):
        f = feats.clone().requires_grad_(True)
        conv.weight.grad = None
        canvas = F.pillar_scatter(f, coors, B, ny, nx, channels_last=True, unique=True, num_valid=nv)
        y = pillar_conv.pillar_conv2d(canvas, conv) if fused else conv(canvas)
        y.square().sum().backward()
        out.append((f.grad.clone(), conv.weight.grad.clone()))
    assert float(out[1][0][250:].abs().max()) == 0.0            # rows past the valid count get no gradient
    assert float((out[0][0] - out[1][0]).abs().max()) <= 2e-5 * float(out[0][0].abs().max())
    assert float((out[0][1] - out[1][1]).abs().max()) <= 2e-5 * float(out[0][1].abs().max())
    # not eligible: NCHW canvas, non-unique scatter, biased conv, no grad
    f = feats.clone().requires_grad_(True)
    assert not pillar_conv.eligible(conv, F.pillar_scatter(f, coors, B, ny, nx, channels_last=False, unique=True))
    assert not pillar_conv.eligible(conv, F.pillar_scatter(f, coors, B, ny, nx, channels_last=True, unique=False))
    cb = torch.nn.Conv2d(C, C, 3, padding=1).to(DEV)
    assert not pillar_conv.eligible(cb, F.pillar_scatter(f, coors, B, ny, nx, channels_last=True, unique=True))
    with torch.no_grad():
        assert not pillar_conv.eligible(conv, F.pillar_scatter(f, coors, B, ny, nx, channels_last=True, unique=True))


# ----------------------------------------------------------------------------- dense 3x3 conv (bf16x9)
@pytest.mark.parametrize('cin,cout,B,H,W', [(64, 64, 2, 37, 45), (128, 128, 1, 19, 70), (128, 64, 2, 8, 32),
                                           (64, 128, 1, 41, 33), (384, 64, 1, 20, 36),
                                           # maps walked transposed (width a poor multiple of 32) and 256-wide outputs in slices
                                           (64, 64, 2, 62, 54), (128, 128, 1, 31, 22), (256, 256, 1, 30, 20),
                                           (128, 256, 1, 9, 40),
                                           # 16-row / 512-thread tiles (cout 128 and >= 384 tiles): straight and transposed walk
                                           (128, 128, 12, 128, 128), (128, 128, 16, 124, 108), (64, 128, 13, 120, 128),
                                           # 256 outputs on 16-row tiles: both slices in one multi-entry launch
                                           (64, 256, 12, 128, 128)])
@pytest.mark.parametrize('planes', [2, 3])
def test_dense_conv3x3_vs_torch(cin, cout, B, H, W, planes, monkeypatch):
    """fp32 convolution through split-plane partial products (planes = 2: two fp16 planes of the scaled operands,
    three products - the default; planes = 3: three bf16 planes, six products): forward, backward-data and weight
    gradient against torch's convolution in float64 (error no larger than a few fp32 ulps of the accumulated sum)."""
    from gga_amd import dense_conv
    monkeypatch.setattr(dense_conv, 'PLANES', planes)
    torch.manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False).to(DEV)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert dense_conv.eligible(conv, x)
    y = dense_conv.conv2d(x, conv)
    assert 'Conv3x3' in type(y.grad_fn).__name__ and y.is_contiguous(memory_format=torch.channels_last)
    xd = x.detach().double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xd, conv.weight.detach().double(), padding=1)
    assert float((y.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    g = torch.randn_like(y)
    y.backward(g)
    ref.backward(g.double())
    assert float((x.grad.double() - xd.grad).abs().max()) <= 3e-6 * float(xd.grad.abs().max())
    gw_ref = torch.nn.grad.conv2d_weight(x.detach().double(), conv.weight.shape, g.double(), padding=1)
    assert float((conv.weight.grad.double() - gw_ref).abs().max()) <= 1e-5 * float(gw_ref.abs().max())   # bf16x9 weight gradient
    # not eligible: stride 2, NCHW input; a bias is added by one elementwise pass after the kernel
    assert not dense_conv.eligible(torch.nn.Conv2d(cin, cout, 3, stride=2, padding=1, bias=False).to(DEV), x)
    assert not dense_conv.eligible(conv, x.detach().contiguous())
    cb = torch.nn.Conv2d(cin, cout, 3, padding=1).to(DEV)
    cb.weight.data = cb.weight.data.contiguous(memory_format=torch.channels_last)
    assert dense_conv.eligible(cb, x)
    xb = x.detach().clone().requires_grad_(True)
    yb = dense_conv.conv2d(xb, cb)
    refb = torch.nn.functional.conv2d(xb.detach().double(), cb.weight.detach().double(), cb.bias.detach().double(), padding=1)
    assert float((yb.double() - refb).abs().max()) <= 3e-6 * float(refb.abs().max())
    yb.sum().backward()
    torch.testing.assert_close(cb.bias.grad, torch.full_like(cb.bias, float(B * H * W)), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('cin,cout,B,H,W', [(64, 64, 2, 37, 45), (128, 128, 1, 19, 70), (64, 128, 1, 41, 33), (128, 128, 1, 31, 22),
                                           (256, 256, 1, 30, 20), (128, 64, 2, 8, 32), (32, 64, 3, 16, 33), (32, 128, 2, 9, 40),
                                           # more tiles than workgroups in the grid: every workgroup walks several tiles
                                           (64, 128, 5, 200, 176), (64, 64, 7, 248, 216)])
def test_dense_conv_producer_consumer_form_equals_the_lock_step_form(cin, cout, B, H, W, monkeypatch):
    """Two fp16 planes: dense_conv_ws.hip (producer / consumer waves, persistent tiles, 128-channel slices as one grid) against
    the lock-step kernel it replaced (GGA_DC_WS=0) - same operand planes, same order of the products: bit-identical outputs
    forward and backward-data; the BatchNorm partial sums cover other tiles (8 / 16 rows swapped) and agree after their fold."""
    from gga_amd import dense_conv
    monkeypatch.setattr(dense_conv, 'PLANES', 2)
    torch.manual_seed(cin * 3 + cout)
    x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    x[:, :, ::3, 1::2] = 0.0                                        # ReLU-like zeros
    w = (torch.randn(cout, cin, 3, 3, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, cout, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    got = {}
    for form in ('1', '0', '16'):
        monkeypatch.setenv('GGA_DC_WS', '0' if form == '0' else '1')
        monkeypatch.setenv('GGA_DC_WS_MFMA', '16' if form == '16' else '32')
        y, st = dense_conv._run(x, w, False, True)
        gx = dense_conv._run(g, w, True, False)[0] if cin % 64 == 0 else y      # (two input chunks: forward only - 32 outputs are not a tile)
        got[form] = (y.clone(), st.sum(0), gx.clone())
    torch.cuda.synchronize()
    assert torch.equal(got['1'][0], got['0'][0]) and torch.equal(got['1'][2], got['0'][2])
    # the 16x16x32 consumer form (round 5; whole quads of chunks: cin % 64 == 0, else the 32x32x16 form ran again): the same
    # planes and products, K = 32 per instruction - another summation order, so close, not identical; against float64 below
    for i in (0, 2):
        assert float((got['16'][i] - got['0'][i]).abs().max()) <= 2e-6 * float(got['0'][i].abs().max())
        if cin % 64 == 0:
            assert not torch.equal(got['16'][i], got['0'][i]), 'the 16x16x32 form did not run'
    torch.testing.assert_close(got['16'][1], got['0'][1], rtol=1e-5, atol=1e-6 * float(got['0'][1].abs().max()))
    # (fp32 lane sums over other pixel groups before the f64 fold)
    torch.testing.assert_close(got['1'][1], got['0'][1], rtol=1e-5, atol=1e-6 * float(got['0'][1].abs().max()))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    for form in ('1', '16'):
        assert float((got[form][0].double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
        torch.testing.assert_close(got[form][1][0], ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-4 * float(ref.abs().sum((0, 2, 3)).max()))
    if cin % 64 == 0:
        refg = torch.nn.grad.conv2d_input(x.shape, w.double(), g.double(), padding=1)
        assert float((got['16'][2].double() - refg).abs().max()) <= 3e-6 * float(refg.abs().max())


def test_dense_weight_gradient_with_an_absmax_per_channel_block():
    """Two fp16 planes: a gradient tensor whose 64-channel blocks differ by 2^-26 in magnitude (the head's 960-channel buffer:
    regression branches beside heat-map branches). Under ONE absmax the small block's weight gradient keeps a few bits; with an
    absmax per block (gga_dense_wgrad3x3_block_amax) every block is as accurate as if it were alone."""
    from gga_amd import dense_conv
    torch.manual_seed(0)
    B, H, W = 2, 40, 48
    x = torch.randn(B, 64, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, 192, H, W, device=DEV)
    g[:, 64:128] *= 2.0 ** -26
    g[:, 128:] *= 2.0 ** -13
    g = g.contiguous(memory_format=torch.channels_last)
    w = torch.empty(192, 64, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, g.double(), padding=1)
    was = dense_conv.PLANES
    dense_conv.PLANES = 2
    try:
        one = dense_conv._wgrad(x, g, w, dense_conv._amax_bits(x), dense_conv._amax_bits(g))
        blocks = torch.cat([dense_conv._amax_bits(g[:, c:c + 64].contiguous(memory_format=torch.channels_last)) for c in (0, 64, 128)])
        per = dense_conv._wgrad(x, g, w, dense_conv._amax_bits(x), blocks, g_per_block=True)
    finally:
        dense_conv.PLANES = was
    err = lambda t, c: float((t[c:c + 64].double() - ref[c:c + 64]).norm() / ref[c:c + 64].norm())
    assert err(one, 0) < 1e-6 and err(one, 64) > 1e-4                    # the tensor-wide scale loses the small block ...
    assert all(err(per, c) < 1e-6 for c in (0, 64, 128)), [err(per, c) for c in (0, 64, 128)]       # ... its own scale does not


@pytest.mark.parametrize('cin,cout,B,H,W', [(64, 64, 3, 37, 45), (64, 128, 2, 37, 45), (128, 128, 2, 31, 22),
                                           (128, 128, 12, 128, 128), (128, 128, 16, 124, 108),
                                           # 256 output channels: both 128-channel slices in one launch, straight and transposed
                                           (128, 256, 2, 37, 45), (256, 256, 4, 62, 54), (64, 256, 12, 128, 128)])
def test_dense_conv_leaves_batchnorm_partials(cin, cout, B, H, W):
    """The statistics epilogue of the dense conv equals the reductions of its output, and the
    BatchNorm fed with them equals the BatchNorm that reduces y itself - 64 and 128 output channels,
    8-row and 16-row tiles, straight and transposed walk."""
    import copy
    from gga_amd import dense_conv
    torch.manual_seed(7)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).to(DEV)
    bn2 = copy.deepcopy(bn)
    x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    y = dense_conv.conv2d(x, conv, bn_follows=True)
    p = y.bn_partials
    assert p.dtype == torch.float64 and p.shape[1:] == (2, cout)
    yd = y.detach().double()
    # per-tile sums are fp32 over 256 pixels, then f64: error relative to the sum of magnitudes
    mag = float(yd.abs().sum((0, 2, 3)).max())
    torch.testing.assert_close(p[:, 0].sum(0), yd.sum((0, 2, 3)), rtol=0, atol=2e-6 * mag)
    torch.testing.assert_close(p[:, 1].sum(0), (yd * yd).sum((0, 2, 3)), rtol=2e-6, atol=0)
    out = F.bn_act(y, bn, relu=True)
    y2 = y.detach().clone()                       # no partials attached: the BatchNorm reduces y itself
    assert getattr(y2, 'bn_partials', None) is None
    ref = F.bn_act(y2, bn2, relu=True)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bn.running_mean, bn2.running_mean, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(bn.running_var, bn2.running_var, rtol=1e-5, atol=1e-7)
    g = torch.randn_like(out)
    out.backward(g)
    assert conv.weight.grad is not None and torch.isfinite(conv.weight.grad).all()


@pytest.mark.parametrize('kind,cin,cout,k,s,B,H,W', [('conv', 64, 128, 3, 2, 2, 37, 45), ('conv', 128, 256, 3, 2, 2, 24, 30),
                                                      ('deconv', 128, 128, 2, 2, 2, 19, 23), ('deconv', 64, 128, 1, 1, 2, 37, 45)])
def test_gather_convs_leave_batchnorm_partials(kind, cin, cout, k, s, B, H, W):
    """The strided / transposed convolutions (gather-GEMM kernel, gga_sparse_conv_apply_stats) hand the BatchNorm that
    follows the per-channel sums of their output: equal to the reductions of y, and the BatchNorm fed with them equals the
    BatchNorm that reduces y itself."""
    import copy
    from gga_amd import strided_conv
    torch.manual_seed(11)
    if kind == 'conv':
        m = torch.nn.Conv2d(cin, cout, k, stride=s, padding=k // 2, bias=False).to(DEV)
    else:
        m = torch.nn.ConvTranspose2d(cin, cout, k, stride=s, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).to(DEV)
    bn2 = copy.deepcopy(bn)
    x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    assert strided_conv.eligible(m, x)
    y = strided_conv.conv(x, m)
    p = y.bn_partials
    assert p.dtype == torch.float64 and p.dim() == 3 and p.shape[1:] == (2, cout)
    yd = y.detach().double()
    mag = float(yd.abs().sum((0, 2, 3)).max())
    torch.testing.assert_close(p[:, 0].sum(0), yd.sum((0, 2, 3)), rtol=0, atol=2e-6 * mag)
    torch.testing.assert_close(p[:, 1].sum(0), (yd * yd).sum((0, 2, 3)), rtol=2e-6, atol=0)
    out = F.bn_act(y, bn, relu=True)
    ref = F.bn_act(y.detach().clone(), bn2, relu=True)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bn.running_var, bn2.running_var, rtol=1e-5, atol=1e-7)
    out.backward(torch.randn_like(out))
    assert m.weight.grad is not None and torch.isfinite(m.weight.grad).all()
    # the sums are only trusted while y holds what the kernel wrote
    with torch.no_grad():
        y2 = strided_conv.conv(x, m)
        assert F.bn_partials_of(y2) is not None
        y2.mul_(2.0)
        assert F.bn_partials_of(y2) is None


def test_channel_sums_vs_torch():
    """gga_column_sums (bias gradients): channels-last maps, row matrices, and the layouts it hands back to torch."""
    torch.manual_seed(3)
    for shape, cl in (((3, 256, 17, 23), True), ((1000, 64), False), ((2, 27, 5, 7), True), ((2, 64, 9, 9), False)):
        x = torch.randn(*shape, device=DEV)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        ref = x.double().sum(tuple(d for d in range(x.dim()) if d != 1))
        got = F.channel_sums(x)
        assert got.shape == ref.shape
        assert float((got.double() - ref).abs().max()) <= 1e-5 * float(x.abs().sum() / x.shape[1]) + 1e-6


@pytest.mark.parametrize('cin,cout', [(256, 256), (64, 64), (128, 64), (256, 128), (64, 128), (128, 256)])
@pytest.mark.parametrize('planes', [2, 3])
def test_dense_conv3x3_over_several_maps_in_one_launch(cin, cout, planes, monkeypatch):
    """gga_dense_conv3x3_levels: one convolution over five maps of different sizes (the FPN levels of a head tower, each
    with its own absmax) - forward, backward-data and the summed weight gradient against torch in float64, and against
    the map-by-map path of this repo."""
    from gga_amd import dense_conv
    monkeypatch.setattr(dense_conv, 'PLANES', planes)
    torch.manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False).to(DEV)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    sizes = [(24, 78), (12, 39), (6, 20), (3, 10), (1, 1)]
    if cout >= 128 and cin <= 128:      # a map with >= 384 16-row tiles: it goes into a launch of its own on the 512-thread form
        sizes = [(256, 256)] + sizes
    xs = [(torch.randn(3, cin, h, w, device=DEV) * 10.0 ** (i - 2)).contiguous(memory_format=torch.channels_last).requires_grad_(True)
          for i, (h, w) in enumerate(sizes)]
    assert dense_conv.levels_eligible(conv, xs)
    ys = dense_conv.conv2d_levels(xs, conv)
    assert all('Conv3x3Levels' in type(y.grad_fn).__name__ for y in ys)
    gs = [torch.randn_like(y) for y in ys]
    torch.autograd.backward(ys, gs)
    gw = conv.weight.grad.clone()
    gw64 = 0
    for x, y, g in zip(xs, ys, gs):
        xd = x.detach().double().requires_grad_(True)
        ref = torch.nn.functional.conv2d(xd, conv.weight.detach().double(), padding=1)
        assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
        assert float((y.detach().double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
        ref.backward(g.double())
        assert float((x.grad.double() - xd.grad).abs().max()) <= 3e-6 * float(xd.grad.abs().max())
        gw64 = gw64 + torch.nn.grad.conv2d_weight(x.detach().double(), conv.weight.shape, g.double(), padding=1)
    assert float((gw.double() - gw64).abs().max()) <= 1e-5 * float(gw64.abs().max())
    # the map-by-map path: same arithmetic, possibly another tile walk
    conv.weight.grad = None
    for x in xs:
        x.grad = None
    ys1 = [dense_conv.conv2d(x, conv) for x in xs]
    for y, y1 in zip(ys, ys1):
        assert float((y - y1).abs().max()) <= 2e-6 * float(y1.abs().max())
    # with a bias: added in the kernel's epilogue, its gradient is the sum of the output gradients
    cb = torch.nn.Conv2d(cin, cout, 3, padding=1).to(DEV)
    cb.weight.data = conv.weight.data.clone()
    assert dense_conv.levels_eligible(cb, xs)
    yb = dense_conv.conv2d_levels(xs, cb)
    for y, y0 in zip(yb, ys):
        assert float((y - (y0.detach() + cb.bias.view(1, -1, 1, 1))).abs().max()) <= 1e-6 * float(y0.abs().max())
    torch.autograd.backward(yb, gs)
    want_gb = sum(g.double().sum((0, 2, 3)) for g in gs)
    assert float((cb.bias.grad.double() - want_gb).abs().max()) <= 1e-6 * float(sum(g.double().abs().sum((0, 2, 3)) for g in gs).max())
    torch.testing.assert_close(cb.weight.grad, gw, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('c0,c1,B,H,W', [(64, 64, 2, 37, 45), (128, 128, 2, 31, 22), (256, 256, 1, 30, 20), (64, 128, 1, 41, 33),
                                         (128, 128, 13, 120, 128), (128, 128, 16, 124, 108)])
@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('form', ['planes3', 'ws32', 'ws16'])
def test_conv_backward_data_reduces_the_batchnorm_below(c0, c1, B, H, W, training, form, monkeypatch):
    """conv -> BatchNorm -> ReLU -> conv: the second convolution's backward-data launch masks its result with the ReLU
    and leaves the BatchNorm backward sums (gga_dense_conv3x3_bn_bwd), and the BatchNorm backward then runs without its
    reduce pass (gga_bn_relu_bwd_partials). Checked against the unfused path of this repo (same kernels otherwise) and
    against torch in float64; 64 / 128 / sliced 256 channels, 8- and 16-row tiles, straight and transposed walk,
    batch and running statistics."""
    import copy
    from gga_amd import dense_conv, _lib
    # the library default (three bf16 planes, lock-step kernel) and the two-plane producer / consumer kernel with either matrix
    # instruction in its consumer waves (its own epilogue code per form)
    if form != 'planes3':
        monkeypatch.setattr(dense_conv, 'PLANES', 2)
        monkeypatch.setenv('GGA_DC_WS', '1')
        monkeypatch.setenv('GGA_DC_WS_MFMA', form[2:])
    torch.manual_seed(c0 + c1 + H)
    conv1 = torch.nn.Conv2d(64, c0, 3, padding=1, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(c0, eps=1e-3, momentum=0.01).to(DEV)
    conv2 = torch.nn.Conv2d(c0, c1, 3, padding=1, bias=False).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.2, 0.2), bn.running_var.uniform_(0.5, 1.5)
    for m in (conv1, conv2):
        m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    bn.train(training)
    x = torch.randn(B, 64, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, c1, H, W, device=DEV).contiguous(memory_format=torch.channels_last)

    L = _lib.lib()
    calls = {'fused': 0}
    real = L.gga_bn_relu_bwd_partials

    def counted(*a):
        calls['fused'] += 1
        return real(*a)

    def run(fused):
        monkeypatch.setattr(dense_conv, 'BN_BWD_FUSED', fused)
        mods = copy.deepcopy((conv1, bn, conv2))
        xi = x.clone().requires_grad_(True)
        y = dense_conv.conv2d(xi, mods[0], bn_follows=True)
        z = F.bn_act(y, mods[1], relu=True)
        out = dense_conv.conv2d(z, mods[2])
        out.backward(g)
        return out.detach(), xi.grad, mods[0].weight.grad, mods[1].weight.grad, mods[1].bias.grad, mods[2].weight.grad

    monkeypatch.setattr(L, 'gga_bn_relu_bwd_partials', counted)
    # the product takes the epilogue only where it is cheaper than the reduce pass; here every tile form is exercised
    pays = L.gga_dense_conv3x3_bn_bwd_pays
    assert pays(16, 248, 216, 64) == 1 and pays(16, 124, 108, 128) == 1 and pays(16, 62, 54, 128) == 0     # lock-step forms
    assert L.gga_dense_conv3x3_bn_bwd_pays_planes(16, 62, 54, 128, 3) == 0
    monkeypatch.setattr(L, 'gga_dense_conv3x3_bn_bwd_pays_planes', lambda *a: 1)
    got = run(True)
    assert calls['fused'] == 1, 'the fused BatchNorm backward did not run'
    plain = run(False)
    assert calls['fused'] == 1
    names = ('out', 'grad x', 'grad conv1', 'grad gamma', 'grad beta', 'grad conv2')
    for n, a, b in zip(names, got, plain):      # same arithmetic, other summation order of the per-channel sums
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7, n
    # float64 torch
    m64 = copy.deepcopy((conv1, bn, conv2))
    for m in m64:
        m.double()
    xd = x.double().requires_grad_(True)
    o = m64[2](torch.relu(m64[1](m64[0](xd))))
    o.backward(g.double())
    ref = (o.detach(), xd.grad, m64[0].weight.grad, m64[1].weight.grad, m64[1].bias.grad, m64[2].weight.grad)
    # L2: fp32 and fp64 disagree on the sign of bn(y) for about one element in a million, and such an element carries its
    # whole gradient (2.7e-4 of the norm at 25 M elements - the unfused path shows the same figure)
    for n, a, b in zip(names, got, ref):
        assert float((a.double() - b).norm()) <= 1e-3 * float(b.norm()) + 1e-9, n


@pytest.mark.parametrize('kind,cin,cout,k,s,B,H,W', [('conv', 64, 128, 3, 2, 2, 37, 45), ('conv', 128, 256, 3, 2, 2, 24, 30),
                                                      ('conv', 64, 64, 3, 2, 1, 40, 32), ('conv', 256, 64, 1, 1, 1, 20, 24),
                                                      ('deconv', 64, 128, 1, 1, 2, 37, 45), ('deconv', 128, 128, 2, 2, 2, 19, 23),
                                                      ('deconv', 256, 128, 4, 4, 2, 9, 11), ('deconv', 256, 256, 2, 2, 1, 8, 8)])
@pytest.mark.parametrize('planes', [2, 3])
def test_strided_and_transposed_convs_vs_torch(kind, cin, cout, k, s, B, H, W, planes, monkeypatch):
    """The SECOND stage openers (3x3 / stride 2) and the SECONDFPN transposed convolutions (kernel = stride) on
    the gather-GEMM kernels (gga_amd/strided_conv.py): output, input gradient and weight gradient against
    float64 torch, next to the framework's own fp32 convolution."""
    from gga_amd import dense_conv, strided_conv
    monkeypatch.setattr(dense_conv, 'PLANES', planes)
    torch.manual_seed(5)
    if kind == 'conv':
        m = torch.nn.Conv2d(cin, cout, k, s, k // 2, bias=False)
    else:
        m = torch.nn.ConvTranspose2d(cin, cout, k, s, bias=False)
    m = m.to(DEV).to(memory_format=torch.channels_last)
    x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert strided_conv.eligible(m, x)
    y = dense_conv.conv2d(x, m)
    assert type(y.grad_fn).__name__ in ('_StridedConvBackward', '_DeconvBackward')
    assert y.is_contiguous(memory_format=torch.channels_last)
    # the per-channel sums the kernel leaves for the BatchNorm that follows (gather-GEMM tiles, or the streaming kernel's
    # per-workgroup sums for the 1x1 / kernel = stride forms): the sums of y and of y^2
    from gga_amd import functional as GF
    parts = GF.bn_partials_of(y)
    assert parts is not None and parts.dtype == torch.float64
    yd = y.detach().double()
    torch.testing.assert_close(parts[:, 0].sum(0), yd.sum((0, 2, 3)), rtol=1e-6, atol=1e-3)
    torch.testing.assert_close(parts[:, 1].sum(0), (yd * yd).sum((0, 2, 3)), rtol=1e-6, atol=1e-3)
    g = torch.randn_like(y)
    y.backward(g)
    gx, gw = x.grad.clone(), m.weight.grad.clone()
    m64 = type(m)(cin, cout, k, s, k // 2 if kind == 'conv' else 0, bias=False).to(DEV).double()
    m64.weight.data.copy_(m.weight.data.double())
    x64 = x.detach().double().requires_grad_(True)
    y64 = m64(x64)
    y64.backward(g.double())
    x.grad = None
    m.weight.grad = None
    y32 = m(x)
    y32.backward(g)
    rel = lambda a, b: float((a.detach().double() - b.detach()).abs().max() / b.detach().abs().max())
    for name, mine, ref, fw in (('y', y, y64, y32), ('gx', gx, x64.grad, x.grad), ('gw', gw, m64.weight.grad, m.weight.grad)):
        assert mine.shape == ref.shape, name
        e, e32 = rel(mine, ref), rel(fw, ref)
        assert e <= max(3e-6, 2 * e32), (name, e, e32)
    # run-to-run identical (no atomics anywhere on this path)
    x.grad = None
    m.weight.grad = None
    y2 = dense_conv.conv2d(x, m)
    y2.backward(g)
    assert torch.equal(y2, y) and torch.equal(x.grad, gx) and torch.equal(m.weight.grad, gw)


@pytest.mark.parametrize('planes', [2, 3])
def test_dense_conv3x3_non_finite_and_wide_range(planes, monkeypatch):
    """Contract of the split-plane arithmetic at the edges of fp32 (include/gga_hip.h).
    planes = 3 (three bf16 planes, six products): finite inputs of any magnitude (2^-100 .. 2^100, denormals) give
    the fp32 result to the usual accumulation error, per output.
    planes = 2 (two fp16 planes of the scaled operands, three products - the default): the same while the magnitudes
    inside a tensor stay within ~2^17 of each other; in general |error| <= 2e-6 * sum|a*b| + 2^-40 * max|x| * sum|w|
    (an element is kept to an absolute accuracy of 2^-39 of its tensor's largest magnitude).
    Both: a non-finite input makes exactly the outputs non-finite that an fp32 convolution makes non-finite (as NaN: Inf
    splits into Inf - Inf), and no other output moves."""
    from gga_amd import dense_conv
    monkeypatch.setattr(dense_conv, 'PLANES', planes)
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(DEV)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    base = torch.randn(2, 64, 24, 40, device=DEV)
    # wide range: every pixel has its own power-of-two scale (a dot product's terms share it, so the
    # float64 reference bounds the error per output by its own magnitude)
    for span in (8, 100):
        e = torch.randint(-span, span + 1, (2, 1, 24, 40), device=DEV).float()
        x = (base * torch.exp2(e)).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y = dense_conv.conv2d(x, conv)
            mag = torch.nn.functional.conv2d(x.double().abs(), conv.weight.double().abs(), padding=1)
            ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), padding=1)
            sum_w = torch.nn.functional.conv2d(torch.ones_like(x).double(), conv.weight.double().abs(), padding=1)
        assert torch.isfinite(y).all()
        err = (y.double() - ref).abs()
        if planes == 3 or span == 8:
            assert float((err / mag).max()) < 5e-6      # relative to sum |a*b| of the same output
        assert bool((err <= 2e-6 * mag + 2.0 ** -40 * float(x.abs().max()) * sum_w).all())
        assert float(err.max()) < 1e-6 * float(ref.abs().max())
    # denormal inputs: result within one denormal-product's worth of the float64 one (flush-to-zero or not)
    xd = (base * 1e-41).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        yd = dense_conv.conv2d(xd, conv)
        refd = torch.nn.functional.conv2d(xd.double(), conv.weight.double(), padding=1)
    assert torch.isfinite(yd).all() and float((yd.double() - refd).abs().max()) < 1e-37
    # non-finite inputs
    xn = base.clone()
    xn[0, 3, 5, 7] = float('inf')
    xn[1, 10, 20, 33] = float('-inf')
    xn[1, 0, 0, 0] = float('nan')
    xn = xn.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        yn = dense_conv.conv2d(xn, conv)
        rn = torch.nn.functional.conv2d(xn.double(), conv.weight.double(), padding=1)
        y0 = dense_conv.conv2d(base.contiguous(memory_format=torch.channels_last), conv)
    bad = ~torch.isfinite(rn)
    assert bad.any() and torch.equal(~torch.isfinite(yn), bad)
    assert torch.equal(yn[~bad], y0[~bad])                          # every other output is bit-identical


@pytest.mark.parametrize('cout', [1, 3])
def test_head_output_conv_weight_gradient_is_repeatable_beside_another_stream(cout):
    """The weight gradient of the head's output convolutions, repeated while a second (high-priority) stream keeps kernels
    running on the same CUs, is the same bits every time. (Round 3 found the kernel staging the first tile's grad_y into LDS
    words that other waves were still zeroing - no barrier between the two: once in ~2000 train steps, and only with the
    next batch's front running beside the step, one tile's gradient was partly wiped: 3e-3 of a branch's weight gradient.)"""
    torch.manual_seed(40 + cout)
    B, H, W = 8, 200, 176
    conv = torch.nn.Conv2d(64, cout, 3, padding=1, bias=True).to(DEV)
    x = torch.randn(B, 64, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, cout, H, W, device=DEV)
    side = torch.cuda.Stream(priority=-1)
    noise = torch.randn(1 << 22, device=DEV)
    keys = torch.randint(0, 1 << 30, (1 << 20,), device=DEV)
    ref = None
    for it in range(40):
        xi = x.clone().requires_grad_(True)
        conv.weight.grad = conv.bias.grad = None
        y = F.head_conv3x3(xi, conv)
        if it >= 5:
            with torch.cuda.stream(side):
                for _ in range(6):
                    noise.mul_(1.0001)
                    torch.sort(keys)
        y.backward(g)
        torch.cuda.synchronize()
        got = (conv.weight.grad.clone(), conv.bias.grad.clone())
        if ref is None:
            ref = got
        else:
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), it


def _repeat_beside_noise(fn, repeats=30):
    """Run ``fn`` (-> tuple of tensors) ``repeats`` times, from the sixth on beside a high-priority stream of small kernels;
    every result must equal the first one bit for bit."""
    side = torch.cuda.Stream(priority=-1)
    noise = torch.randn(1 << 22, device=DEV)
    keys = torch.randint(0, 1 << 30, (1 << 20,), device=DEV)
    ref = None
    for it in range(repeats):
        if it >= 5:
            with torch.cuda.stream(side):
                for _ in range(6):
                    noise.mul_(1.0001)
                    torch.sort(keys)
                    torch.cumsum(noise, 0)
        got = [t.clone() for t in fn()]
        torch.cuda.synchronize()
        if ref is None:
            ref = got
        else:
            for a, b in zip(got, ref):
                assert torch.equal(a, b), it


@pytest.mark.parametrize('planes', [2, 3])
def test_matrix_and_norm_kernels_are_repeatable_beside_another_stream(planes, monkeypatch):
    """The kernels of the camera-only step that the LiDAR twins do not reach, and the dense / strided convolutions at other
    shapes, each repeated beside a busy second stream: fused GroupNorm + ReLU forward / backward, the dense 3x3 convolution with
    256 -> 256 channels (two output slices per launch) forward / backward / weight gradient, a stride-2 3x3 and a 1x1
    convolution on the gather-GEMM kernels. Bit-identical every time (none of them adds with float atomics)."""
    from gga_amd import dense_conv, strided_conv
    monkeypatch.setattr(dense_conv, 'PLANES', planes)
    torch.manual_seed(5)
    x = torch.randn(4, 256, 48, 156, device=DEV).contiguous(memory_format=torch.channels_last)
    gn = torch.nn.GroupNorm(32, 256).to(DEV)
    conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(DEV)
    down = torch.nn.Conv2d(256, 128, 3, stride=2, padding=1, bias=False).to(DEV)
    one = torch.nn.Conv2d(256, 64, 1, bias=False).to(DEV)
    g = torch.randn(4, 256, 48, 156, device=DEV).contiguous(memory_format=torch.channels_last)

    def gn_pass():
        xi = x.clone().requires_grad_(True)
        gn.weight.grad = gn.bias.grad = None
        y = F.gn_act(xi, gn, relu=True)
        y.backward(g)
        return y.detach(), xi.grad, gn.weight.grad, gn.bias.grad

    def conv_pass(m):
        def run():
            xi = x.clone().requires_grad_(True)
            m.weight.grad = None
            y = dense_conv.conv2d(xi, m) if m is conv else strided_conv.conv(xi, m)
            y.backward(torch.ones_like(y) * 0.5 + y.detach() * 0.1)
            return y.detach(), xi.grad, m.weight.grad
        return run

    assert dense_conv.eligible(conv, x) and strided_conv.eligible(down, x) and strided_conv.eligible(one, x)
    _repeat_beside_noise(gn_pass)
    for m in (conv, down, one):
        _repeat_beside_noise(conv_pass(m), repeats=20)


@pytest.mark.parametrize('planes', [2, 3])
def test_weight_bank_operands_equal_the_per_call_packing(planes):
    """gga_pack_weights_table / gga_absmax_table (the weight bank's two launches per step) against the per-convolution kernels
    they replace, bit for bit: dense 3x3 operands (forward / backward-data, both tile walks, 128-channel slices of a wide
    weight, channels-last and contiguous parameters), operands assembled from several weights against the packed torch.cat,
    gather-GEMM operands from permuted views against the packed reshaped copy; a changed weight refreshes every operand."""
    from gga_amd import _lib, dense_conv, weight_bank as WB
    from gga_amd._lib import check
    L = _lib.lib()
    torch.manual_seed(planes)

    def old_dense(w, backward, transposed):
        cout, cin = w.shape[0], w.shape[1]
        n_in, n_out = (cout, cin) if backward else (cin, cout)
        wp = torch.zeros(L.gga_sparse_split_weight_bytes(9, n_in, n_out) // 2, dtype=torch.int16, device=DEV)
        amax = dense_conv._amax_bits(w) if planes == 2 else None
        s = w.stride()
        sky, skx = (s[3], s[2]) if transposed else (s[2], s[3])
        check(L.gga_dense_conv3x3_pack_planes(F._p(w), s[0], s[1], sky, skx, cin, cout, int(backward), planes, F._p(amax), F._p(wp),
                                              F._stream()), 'pack')
        return wp, amax

    bank = WB.WeightBank()
    ws = [torch.randn(64, 64, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last) * (0.5 + i) for i in range(3)]
    wide = torch.randn(256, 128, 3, 3, device=DEV)                       # contiguous (NCHW) parameter
    odd = torch.randn(64, 96, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)       # 96 inputs: a half-empty last stage
    old_bank, WB.BANK = WB.BANK, bank
    try:
        def check_all():
            for w in (ws[0], odd):
                for backward in (False, True):
                    for tr in (False, True):
                        got, slot = WB.dense_operand(w, backward, tr, planes)
                        want, amax = old_dense(w, backward, tr)
                        assert torch.equal(got, want), (tuple(w.shape), backward, tr)
                        assert planes == 3 or torch.equal(slot, amax)
            for backward in (False, True):                                   # slices of a wide weight share the whole weight's scale
                n_out = wide.shape[1] if backward else wide.shape[0]
                for c0 in range(0, n_out, 128):
                    got, slot = WB.dense_operand(wide, backward, False, planes, c0)
                    sl = wide[:, c0:c0 + 128] if backward else wide[c0:c0 + 128]
                    cout, cin = sl.shape[0], sl.shape[1]
                    n_in, n_o = (cout, cin) if backward else (cin, cout)
                    want = torch.zeros(L.gga_sparse_split_weight_bytes(9, n_in, n_o) // 2, dtype=torch.int16, device=DEV)
                    amax = dense_conv._amax_bits(wide) if planes == 2 else None
                    s = sl.stride()
                    check(L.gga_dense_conv3x3_pack_planes(F._p(sl), s[0], s[1], s[2], s[3], cin, cout, int(backward), planes, F._p(amax),
                                                          F._p(want), F._stream()), 'pack')
                    assert torch.equal(got, want) and (planes == 3 or torch.equal(slot, amax)), (backward, c0)
            # virtual concatenations: two weights side by side (forward), three one after the other (backward-data)
            got, slot = WB.dense_operand_cat(ws[:2], False, False, planes)
            want, amax = old_dense(torch.cat(ws[:2], 0).contiguous(memory_format=torch.channels_last), False, False)
            assert torch.equal(got, want) and (planes == 3 or torch.equal(slot, amax))
            got, slot = WB.dense_operand_cat(ws, True, True, planes)
            want, amax = old_dense(torch.cat(ws, 0).contiguous(memory_format=torch.channels_last), True, True)
            assert torch.equal(got, want) and (planes == 3 or torch.equal(slot, amax))
            # gather-GEMM operands: Conv2d weight [cout, cin, k, k] as [k, k, cin, cout] (forward) and [k, k, cout, cin] (backward-data)
            cw = torch.randn(128, 64, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
            for view, n_in, n_out in ((cw.permute(2, 3, 1, 0), 64, 128), (cw.permute(2, 3, 0, 1), 128, 64)):
                got, slot = WB.gather_operand(view, planes)
                flat = view.reshape(9, n_in, n_out).contiguous()
                want = torch.zeros(L.gga_sparse_split_weight_bytes(9, n_in, n_out) // 2, dtype=torch.int16, device=DEV)
                amax = dense_conv._amax_bits(cw) if planes == 2 else None
                check(L.gga_sparse_pack_weight_planes(F._p(flat), 9, n_in, n_out, 0, planes, F._p(amax), F._p(want), F._stream()), 'pack')
                assert torch.equal(got, want) and (planes == 3 or torch.equal(slot, amax))
        check_all()
        n_ops, launches = len(bank.ops), bank.refreshes
        assert n_ops >= 15
        with torch.no_grad():                                               # an optimizer step: every weight changes in place
            for w in ws + [wide, odd]:
                w.mul_(1.7).add_(0.01)
        check_all()
        assert len(bank.ops) == n_ops + 2 and bank.generation == 1          # (+ the fresh cw operands of the second pass)
        assert bank.refreshes == launches + 1 + 2                           # ONE refresh for all stale operands (+ the two new ones)
        # torch's fused optimizers update the parameters without touching their version counters: the step hook that
        # train.build_optimizer registers (weight_bank.BANK.invalidate) is what tells the bank
        params = [torch.nn.Parameter(w) for w in ws + [wide, odd]]          # (share the tensors' memory)
        opt = torch.optim.AdamW(params, lr=0.1, fused=True)
        opt.register_step_post_hook(lambda *_: bank.invalidate())
        for p in params:
            p.grad = torch.randn_like(p)
        before = ws[0].clone()
        opt.step()
        assert not torch.equal(ws[0], before)
        check_all()
        assert bank.generation == 2
    finally:
        WB.BANK = old_bank
