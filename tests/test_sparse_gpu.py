"""Sparse 3D convolution (a3') on the GPU against the dense-conv3d restatement
(oracle/sparse_ref.py): single layers (forward, backward-data, backward-weight), output-site
sets, and the whole SparseEncoder of the GGA config on a small grid."""
import copy

import numpy as np
import pytest
import torch

from gga_amd.sparse import SparseConv3d, SparseConvTensor, SubMConv3d
from gga_amd.sparse_encoder import SparseEncoder
from oracle import sparse_ref as SR

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _coords(batch, shape, n, seed, clustered=True):
    g = torch.Generator().manual_seed(seed)
    D, H, W = shape
    out = []
    for b in range(batch):
        if clustered:       # blobs, so that neighbours exist
            ctr = torch.stack([torch.randint(0, D, (8,), generator=g), torch.randint(0, H, (8,), generator=g),
                               torch.randint(0, W, (8,), generator=g)], 1)
            pts = ctr[torch.randint(0, 8, (n,), generator=g)] + torch.randint(-3, 4, (n, 3), generator=g)
        else:
            pts = torch.stack([torch.randint(0, D, (n,), generator=g), torch.randint(0, H, (n,), generator=g),
                               torch.randint(0, W, (n,), generator=g)], 1)
        pts[:, 0].clamp_(0, D - 1), pts[:, 1].clamp_(0, H - 1), pts[:, 2].clamp_(0, W - 1)
        pts = torch.unique(pts, dim=0)
        pts = pts[torch.randperm(len(pts), generator=g)]
        out.append(torch.cat([torch.full((len(pts), 1), b), pts], 1))
    return torch.cat(out).int()


def _sorted_rows(feats, coors, shape):
    D, H, W = shape
    c = coors.long()
    key = ((c[:, 0] * D + c[:, 1]) * H + c[:, 2]) * W + c[:, 3]
    order = torch.argsort(key)
    return feats[order], key[order]


@pytest.mark.parametrize('cfg', [
    dict(cls=SubMConv3d, cin=4, cout=16, k=3, s=1, p=1),
    dict(cls=SubMConv3d, cin=32, cout=32, k=3, s=1, p=1),
    dict(cls=SubMConv3d, cin=128, cout=128, k=3, s=1, p=1),
    dict(cls=SparseConv3d, cin=16, cout=32, k=3, s=2, p=1),
    dict(cls=SparseConv3d, cin=64, cout=128, k=3, s=2, p=(0, 1, 1)),
    dict(cls=SparseConv3d, cin=128, cout=128, k=(3, 1, 1), s=(2, 1, 1), p=0),
    dict(cls=SparseConv3d, cin=5, cout=24, k=3, s=1, p=0),
    dict(cls=SubMConv3d, cin=32, cout=64, k=3, s=1, p=1),            # weight-gradient tile shapes (1,2) ...
    dict(cls=SparseConv3d, cin=96, cout=48, k=3, s=2, p=1),          # ... and a padded (4,4)
    dict(cls=SubMConv3d, cin=64, cout=64, k=3, s=1, p=1),            # (2,2)
])
@pytest.mark.parametrize('planes', [2, 3])       # two fp16 planes / three products (default), three bf16 planes / six
def test_single_conv_fwd_bwd(cfg, planes, monkeypatch):
    from gga_amd import dense_conv
    monkeypatch.setattr(dense_conv, 'PLANES', planes)
    torch.manual_seed(0)
    shape, B = (11, 24, 20), 2
    coors = _coords(B, shape, 300, seed=1)
    feats = torch.randn(len(coors), cfg['cin'])
    conv = cfg['cls'](cfg['cin'], cfg['cout'], cfg['k'], stride=cfg['s'], padding=cfg['p'], bias=False)
    ref_conv = copy.deepcopy(conv)
    xr = feats.clone().requires_grad_(True)
    yr, cr, shr = SR.conv_ref(ref_conv, xr, coors, B, shape)
    conv.to(DEV)
    xg = feats.to(DEV).requires_grad_(True)
    out = conv(SparseConvTensor(xg, coors.to(DEV), shape, B))
    assert tuple(out.spatial_shape) == tuple(shr)
    ys, ks = _sorted_rows(out.features.detach().cpu(), out.indices.cpu(), shr)
    yrs, krs = _sorted_rows(yr.detach(), cr, shr)
    assert torch.equal(ks, krs)                       # identical output site set (integer work: exact)
    torch.testing.assert_close(ys, yrs, rtol=1e-4, atol=1e-4)
    # backward with a gradient defined per output CELL (orders differ between the two)
    gmap = torch.randn(B, cfg['cout'], *shr)
    def g_at(c):
        c = c.long()
        return gmap[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]]
    yr.backward(g_at(cr))
    out.features.backward(g_at(out.indices.cpu()).to(DEV))
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(conv.weight.grad.cpu(), ref_conv.weight.grad, rtol=1e-4, atol=2e-4)
    # the weight gradient is deterministic (ordered pair compaction, fixed-order reduction, no atomics)
    g1 = conv.weight.grad.clone()
    conv.weight.grad = None
    xg.grad = None
    out2 = conv(SparseConvTensor(xg, coors.to(DEV), shape, B))
    out2.features.backward(g_at(out2.indices.cpu()).to(DEV))
    assert torch.equal(conv.weight.grad, g1)


def test_dense_and_replace_feature():
    shape, B = (5, 12, 8), 3
    coors = _coords(B, shape, 60, seed=3, clustered=False)
    feats = torch.randn(len(coors), 16)
    x = SparseConvTensor(feats.to(DEV), coors.to(DEV), shape, B)
    d = x.dense().cpu()
    assert torch.equal(d, SR._dense(feats, coors, B, shape))
    y = x.replace_feature(x.features * 2)
    assert y._level is x._level and torch.equal(y.dense().cpu(), d * 2)


def _relu_decisions_that_differ(enc, trace, feats, coors, B, y, y64):
    """Number of post-ReLU elements (the inputs of every convolution after the first, and the encoder's output) that
    are live on one side and dead on the other. Such an element sat within rounding of zero before the ReLU: the forward
    values agree, but its gradient is passed on one side and stopped on the other - a kink of the function, not an error
    of either side. On the small grids of these tests one such element moves a layer's gradient by ~1e-2."""
    n = int(((y.detach().cpu() > 0) != (y64.detach() > 0)).sum())
    for (f1, c1, s1), (f2, c2, s2) in list(zip(enc._conv_inputs, trace.inputs))[1:]:
        a, _ = _sorted_rows(f1, c1, s1)
        b, _ = _sorted_rows(f2.detach(), c2, s2)
        n += int(((a > 0) != (b > 0)).sum())
    return n


@pytest.mark.parametrize('planes', [2, 3])
def test_sparse_encoder_gga_config_vs_dense_reference(planes, monkeypatch):
    from gga_amd import dense_conv
    from gga_amd.sparse import SparseConvolution
    from oracle import torch_ref as R
    monkeypatch.setattr(dense_conv, 'PLANES', planes)
    shape, B = (41, 40, 32), 2
    clean = 0
    for seed in range(4):
        torch.manual_seed(seed)
        enc = SparseEncoder(in_channels=4, sparse_shape=list(shape), output_channels=128, order=('conv', 'norm', 'act'),
                            encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                            encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')
        enc.train()
        ref = copy.deepcopy(enc)
        ref64 = copy.deepcopy(enc).double()
        coors = _coords(B, shape, 900, seed=5 + seed)
        feats = torch.randn(len(coors), 4)
        yr, _ = SR.sparse_encoder_reference(ref, feats, coors, B)
        enc.to(DEV)
        enc._conv_inputs = []
        hooks = [m.register_forward_pre_hook(lambda mod, inp: enc._conv_inputs.append(
            (inp[0].features.detach().cpu(), inp[0].indices.cpu(), tuple(inp[0].spatial_shape))))
            for m in enc.modules() if isinstance(m, SparseConvolution)]
        y = enc(feats.to(DEV), coors.to(DEV), B)
        for h in hooks:
            h.remove()
        assert y.shape == yr.shape == (B, 256, 5, 4)
        torch.testing.assert_close(y.detach().cpu(), yr.detach(), rtol=1e-3, atol=1e-3)
        g = torch.randn_like(yr)
        yr.backward(g)
        y.backward(g.to(DEV))
        for (n1, b1), (n2, b2) in zip(enc.named_buffers(), ref.named_buffers()):
            torch.testing.assert_close(b1.cpu(), b2, rtol=1e-4, atol=1e-5, msg=n1)
        # 21 fp32 conv + BN layers deep and the last levels hold few sites: two fp32 implementations differ by more than 1e-3
        # in the first layers' gradients, so each is measured against the same encoder in float64
        # (oracle/torch_ref.gradient_offenders: within 1e-3 of float64, or no further from it than twice the fp32 restatement)
        trace = SR.Trace()
        y64, _ = SR.sparse_encoder_reference(ref64, feats.double(), coors, B, trace=trace)
        y64.backward(g.double())
        if _relu_decisions_that_differ(enc, trace, feats, coors, B, y, y64):
            continue            # a ReLU decision within rounding of zero: this draw cannot be compared (see the helper)
        clean += 1
        grads = {n: p.grad.cpu() for n, p in enc.named_parameters()}
        assert R.gradient_offenders(grads, ref, ref64, tol=1e-3, slack=2.0) == [], seed
    assert clean >= 1, 'every one of four draws had a ReLU decision within rounding of zero'


def test_reference_shape_test():
    # the reference's own (shape-only) sparse-encoder test: tests/test_models/test_common_modules/
    # test_middle_encoders.py:8-27, with unique coordinates
    enc = SparseEncoder(in_channels=5, sparse_shape=[40, 1024, 1024], order=('conv', 'norm', 'act'),
                        encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                        encoder_paddings=((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
                        block_type='basicblock').to(DEV)
    coors = _coords(4, (40, 1024, 1024), 50000, seed=9, clustered=False).to(DEV)
    ret = enc(torch.rand(len(coors), 5, device=DEV), coors, 4)
    assert ret.shape == torch.Size([4, 256, 128, 128])


@pytest.mark.parametrize('planes', [2, 3])
def test_sparse_encoder_full_grid_vs_pair_list_reference(planes, monkeypatch):
    """The SparseEncoder of configs/gga/gga_kitti_config.py at its real grid (41 x 1600 x 1408) on four
    voxelized synthetic KITTI frames (> 50 k input sites): after EVERY convolution the site set
    equals the pair-list restatement's exactly and the features agree; then the gradients of every
    parameter. (The dense conv3d restatement cannot hold this grid; oracle/sparse_ref.conv_ref_pairs
    is pinned to it on small grids by tests/test_oracle.py.)"""
    import os
    from conftest import REPO
    from gga_amd import Config, synthetic
    from gga_amd import functional as F
    from gga_amd.registry import build_middle_encoder
    from gga_amd.sparse import SparseConvolution
    from gga_amd import dense_conv
    monkeypatch.setattr(dense_conv, 'PLANES', planes)
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
    torch.manual_seed(0)
    enc = build_middle_encoder(cfg.model.pts_middle_encoder)
    enc.train()
    ref = copy.deepcopy(enc)
    ref64 = copy.deepcopy(enc).double()
    B = 4
    batch = synthetic.make_batch(B, n_points=20000, pc_range=synthetic.RANGE_SECOND)
    vl = cfg.model.pts_voxel_layer
    v, n, c, _ = F.hard_voxelize_batch([p.to(DEV) for p in batch['points']], vl.voxel_size, vl.point_cloud_range,
                                       vl.max_num_points, vl.max_voxels[0])
    feats = F.voxel_mean(v, n, 4)
    assert len(c) > 50000
    trace = []
    # the restatement is plain torch (index arithmetic, matmul, index_add): it runs on the device too, where the
    # float64 pass over this grid takes seconds instead of minutes; nothing of libgga_hip is involved in it
    ref.to(DEV), ref64.to(DEV)
    yr, _ = SR.sparse_encoder_reference(ref, feats, c, B, pairs=True, trace=trace)
    enc.to(DEV)
    got = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: got.append((out.features.detach().cpu(), out.indices.cpu(),
                                                                       tuple(out.spatial_shape))))
             for m in enc.modules() if isinstance(m, SparseConvolution)]
    y = enc(feats, c, B)
    for h in hooks:
        h.remove()
    assert len(got) == len(trace) == 21
    for i, ((f1, c1, s1), (f2, c2, s2)) in enumerate(zip(got, trace)):
        assert tuple(s1) == tuple(s2), i
        a, ka = _sorted_rows(f1, c1, s1)
        b, kb = _sorted_rows(f2.detach().cpu(), c2.cpu(), s2)
        assert torch.equal(ka, kb), f'conv {i}: site sets differ ({len(ka)} vs {len(kb)})'
        err = float((a - b).norm() / b.norm())
        assert err < 2e-4, (i, len(ka), err)
    assert y.shape == yr.shape == (B, 256, 200, 176)
    torch.testing.assert_close(y.detach(), yr.detach(), rtol=1e-3, atol=1e-3)
    g = torch.randn_like(yr)
    yr.backward(g)
    # every convolution's backward IN SITU (round 6): the kernels' (gx, gw) against float64 from the very operands the call received
    # (features, weight, output gradient) - a deterministic, tight check of the arithmetic that does not depend on which ReLU
    # decisions upstream happened to fall which way (see below)
    from gga_amd import sparse as _sp
    real_bwd, insitu = _sp._SparseConvFn.backward, []

    def bwd(ctx, gy, _gstats=None):
        out = real_bwd(ctx, gy, _gstats)
        f, w = ctx.saved_tensors
        insitu.append((f.detach(), w.detach(), gy.detach().clone(), None if out[0] is None else out[0].detach().clone(),
                       out[1].detach().clone(), ctx.rb.nbr))
        return out
    monkeypatch.setattr(_sp._SparseConvFn, 'backward', staticmethod(bwd))
    y.backward(g)
    monkeypatch.setattr(_sp._SparseConvFn, 'backward', real_bwd)
    assert len(insitu) == 21
    rel = lambda a, b: float((a.double() - b).norm() / b.norm().clamp_min(1e-300))
    worst = 0.0
    for f, w, gy, gx, gw, nbr in insitu:
        nbr = nbr.long()
        x64, w64 = f.double().requires_grad_(True), w.double().requires_grad_(True)
        yy = x64.new_zeros(nbr.shape[1], w64.shape[-1])
        for k in range(nbr.shape[0]):
            rows = (nbr[k] >= 0).nonzero()[:, 0]
            if len(rows):
                yy = yy.index_add(0, rows, x64[nbr[k, rows]] @ w64[k])
        yy.backward(gy.double())
        e_w = rel(gw, w64.grad)
        e_x = 0.0
        if gx is not None:
            e_x = rel(gx, x64.grad)
            if e_x > 1e-3:                 # the launch carried the BatchNorm-backward epilogue: gx is masked by the ReLU below, (f > 0)
                e_x = rel(gx, x64.grad * (f > 0))
        worst = max(worst, e_w, e_x)
        assert e_w < 3e-6 and e_x < 3e-6, (tuple(w.shape), e_x, e_w)
    print(f'FULL_GRID_INSITU planes {planes}: 21 convolutions, backward-data and weight gradient vs float64 from the same operands: worst relative error {worst:.2e}')
    # gradients against the same encoder in float64 (oracle/torch_ref.gradient_offenders: within 1e-3 of
    # float64, or no further from it than twice the fp32 restatement is)
    from oracle import torch_ref as R
    y64, _ = SR.sparse_encoder_reference(ref64, feats.double(), c, B, pairs=True)
    y64.backward(g.double())
    grads = {n: p.grad for n, p in enc.named_parameters()}
    bad = R.gradient_offenders(grads, ref, ref64, tol=1e-3, slack=2.0)
    rows = R.gradient_offenders(grads, ref, ref64, tol=-1.0, slack=0.0)             # every parameter: (name, error, the fp32 restatement's error)
    ratios = sorted(e / max(f, 1e-12) for _, e, f in rows)
    print(f'FULL_GRID_GRADS planes {planes}: {len(rows)} parameters, error vs float64 / the fp32 restatement\'s: median {ratios[len(ratios) // 2]:.2f}, '
          f'90th percentile {ratios[int(len(ratios) * 0.9)]:.2f}, max {ratios[-1]:.2f}; largest error {max(e for _, e, _ in rows):.2e}')
    print('FULL_GRID_ROWS', planes, [(n_.replace('encoder_layers.encoder_layer', 'L'), round(e, 5), round(f, 5)) for n_, e, f in rows])
    print('FULL_GRID_OFFENDERS', planes, bad)
    # End to end against float64 the figure is a property of the ReLU decisions, not of the arithmetic: a pre-activation within
    # rounding of zero passes its gradient in one run and not in the other, and with a random output gradient the effect of a few
    # such elements at the 128-channel level travels to every parameter upstream unchanged (measured round 6, planes 2: one
    # accumulator chain per row - every parameter 3e-4 .. 6e-4 from float64; a chain per offset, whose forward AND backward are
    # 2-3 x closer to float64 per convolution (in situ above, EXPERIMENTS.md 6f) - 1e-6 at the last block, then 3e-3 from
    # encoder_layer4.0.conv1 upstream, where the fp32 restatement is 1e-3 .. 2.6e-3; the whole-step cases of tests/test_model_gpu.py
    # scatter the same way per seed, in both directions). Bound: every parameter within 1e-2, at most a quarter of them beyond
    # the strict criterion (1e-3 / twice the fp32 restatement).
    assert max(e for _, e, _ in rows) < 1e-2 and len(bad) <= 0.25 * len(rows), bad


def test_sparse_conv_leaves_batchnorm_partials():
    """A sparse convolution hands the BatchNorm1d that follows the per-channel sums of its output features
    (gga_sparse_conv_apply_stats): they equal the column sums, and bn_act with them equals bn_act without."""
    import copy
    from gga_amd import functional as F, sparse
    torch.manual_seed(5)
    dev = 'cuda:0'
    shape, B = (9, 40, 36), 2
    coors = torch.unique(torch.stack([torch.randint(0, B, (3000,)), torch.randint(0, shape[0], (3000,)),
                                      torch.randint(0, shape[1], (3000,)), torch.randint(0, shape[2], (3000,))], 1), dim=0).int().to(dev)
    feats = torch.randn(coors.shape[0], 16, device=dev)
    x = sparse.SparseConvTensor(feats, coors, shape, B)
    for conv in (sparse.SubMConv3d(16, 32, 3, padding=1, bias=False, indice_key='s').to(dev),
                 sparse.SparseConv3d(16, 64, 3, stride=2, padding=1, bias=False).to(dev)):
        y = conv(x).features
        p = y.bn_partials
        assert p.dtype == torch.float64 and p.shape[1:] == (2, y.shape[1])
        yd = y.detach().double()
        torch.testing.assert_close(p[:, 0].sum(0), yd.sum(0), rtol=0, atol=2e-6 * float(yd.abs().sum(0).max()))
        torch.testing.assert_close(p[:, 1].sum(0), (yd * yd).sum(0), rtol=2e-6, atol=0)
        bn = torch.nn.BatchNorm1d(y.shape[1], eps=1e-3, momentum=0.01).to(dev)
        bn2 = copy.deepcopy(bn)
        out = F.bn_act(y, bn, relu=True)
        ref = F.bn_act(y.detach().clone(), bn2, relu=True)
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(bn.running_mean, bn2.running_mean, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('mid', [32, 64, 128])       # one, two and four 32-column tiles per wave in the epilogue
@pytest.mark.parametrize('training', [True, False])
def test_sparse_backward_data_reduces_the_batchnorm_below(training, mid, monkeypatch):
    """SubMConv3d -> BatchNorm1d -> ReLU -> SparseConv3d / SubMConv3d: the second convolution's backward-data launch masks
    its result with the ReLU and leaves the BatchNorm backward sums (gga_sparse_conv_apply_bn_bwd), the BatchNorm backward
    runs without its reduce pass. Against the unfused path of this repo (same kernels otherwise)."""
    import copy
    from gga_amd import _lib, dense_conv, functional as F, sparse
    torch.manual_seed(9)
    dev = 'cuda:0'
    shape, B = (9, 40, 36), 2
    coors = torch.unique(torch.stack([torch.randint(0, B, (4000,)), torch.randint(0, shape[0], (4000,)),
                                      torch.randint(0, shape[1], (4000,)), torch.randint(0, shape[2], (4000,))], 1), dim=0).int().to(dev)
    feats = torch.randn(coors.shape[0], 16, device=dev)
    L = _lib.lib()
    calls = {'fused': 0}
    real = L.gga_bn_relu_bwd_partials

    def counted(*a):
        calls['fused'] += 1
        return real(*a)

    monkeypatch.setattr(L, 'gga_bn_relu_bwd_partials', counted)
    for second in (sparse.SubMConv3d(mid, 64, 3, padding=1, bias=False, indice_key='s').to(dev),
                   sparse.SparseConv3d(mid, 128, 3, stride=2, padding=1, bias=False).to(dev)):
        conv1 = sparse.SubMConv3d(16, mid, 3, padding=1, bias=False, indice_key='s').to(dev)
        bn = torch.nn.BatchNorm1d(mid, eps=1e-3, momentum=0.01).to(dev)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5), bn.running_mean.uniform_(-0.2, 0.2), bn.running_var.uniform_(0.5, 1.5)
        bn.train(training)

        def run(fused):
            monkeypatch.setattr(dense_conv, 'BN_BWD_FUSED', fused)
            mods = copy.deepcopy((conv1, bn, second))
            f = feats.clone().requires_grad_(True)
            x = sparse.SparseConvTensor(f, coors, shape, B)
            h = mods[0](x)
            h = h.replace_feature(F.bn_act(h.features, mods[1], relu=True))
            out = mods[2](h).features
            g = torch.linspace(-1, 1, out.numel(), device=dev).view_as(out)
            out.backward(g)
            return out.detach(), f.grad, mods[0].weight.grad, mods[1].weight.grad, mods[1].bias.grad, mods[2].weight.grad

        before = calls['fused']
        got = run(True)
        assert calls['fused'] == before + 1, 'the fused BatchNorm backward did not run'
        plain = run(False)
        assert calls['fused'] == before + 1
        for n, a, b in zip(('out', 'grad feats', 'grad conv1', 'grad gamma', 'grad beta', 'grad conv2'), got, plain):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7, n


@pytest.mark.parametrize('mid', [64, 128])
def test_halo_form_of_the_submanifold_convolution(mid, monkeypatch):
    """GGA_SP_HALO: the submanifold convolutions of 64 / 128 columns through sp_conv_halo_kernel (spatial 256-row tiles whose
    distinct neighbour rows are staged once per 32-channel chunk in LDS) against the default gather-GEMM kernel of this repo -
    forward with the BatchNorm sums, backward-data with the BatchNorm-backward epilogue, on a level dense enough that some
    tiles' halos exceed the 512-row image (the lanes that then read from global memory) and with a last tile that is not full."""
    import copy
    from gga_amd import _lib, dense_conv, functional as F, sparse
    monkeypatch.setattr(dense_conv, 'PLANES', 2)
    torch.manual_seed(11)
    dev = 'cuda:0'
    shape, B = (24, 40, 40), 2
    dense = torch.rand(B, *shape) < 0.45
    coors = dense.nonzero().int()
    coors = coors[torch.randperm(len(coors))].contiguous().to(dev)
    n = coors.shape[0]
    assert n % 256 != 0
    feats = torch.randn(n, 16, device=dev)
    L = _lib.lib()
    calls = {'halo': 0}
    real = L.gga_sparse_conv_apply_halo

    def counted(*a):
        calls['halo'] += 1
        return real(*a)

    monkeypatch.setattr(L, 'gga_sparse_conv_apply_halo', counted)
    conv1 = sparse.SubMConv3d(16, mid, 3, padding=1, bias=False, indice_key='s').to(dev)
    bn = torch.nn.BatchNorm1d(mid, eps=1e-3, momentum=0.01).to(dev)
    conv2 = sparse.SubMConv3d(mid, mid, 3, padding=1, bias=False, indice_key='s').to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5)

    def run(halo):
        monkeypatch.setattr(sparse, 'HALO', halo)
        monkeypatch.setattr(sparse, 'HALO_COLUMNS', (64, 128))       # (the product sends 128 columns only; the kernel takes both)
        monkeypatch.setattr(sparse, 'HALO_MIN_ROWS', 0)
        monkeypatch.setattr(sparse, 'HALO_MIN_OCCUPANCY', 0.0)
        mods = copy.deepcopy((conv1, bn, conv2))
        f = feats.clone().requires_grad_(True)
        x = sparse.SparseConvTensor(f, coors, shape, B)
        h = mods[0](x)
        h = h.replace_feature(F.bn_act(h.features, mods[1], relu=True))
        out = mods[2](h).features        # (no ReLU behind the convolution under test: its decisions would differ at elements near 0)
        sums = out.bn_partials.sum(0)
        g = torch.linspace(-1, 1, out.numel(), device=dev).view_as(out)
        out.backward(g)
        rb = h._level.subm_rulebook((3, 3, 3))
        return (out.detach(), sums, f.grad, mods[0].weight.grad, mods[1].weight.grad, mods[1].bias.grad, mods[2].weight.grad), rb

    got, rb = run(1)
    assert calls['halo'] == 2, 'forward and backward-data of the second convolution did not take the halo form'
    counts = rb.halo().counts
    assert int(counts.max()) > 512 and int(counts.min()) >= 1, (int(counts.min()), int(counts.max()))
    # the tiling itself (gga_sparse_halo_build): every tile's halo holds distinct rows, exactly those its rule-book entries
    # name, and every entry points at its own neighbour
    hl = rb.halo()
    nbr, T = rb.nbr.long(), hl.n_tiles
    rows = hl.tile_rows.long().view(T, 256)
    ent = torch.where(rows[:, None, :] >= 0, nbr[:, rows.clamp(min=0)].permute(1, 0, 2), torch.full((1,), -1, device=dev, dtype=torch.long))
    lm = hl.local_map.long() & 0xFFFF
    assert bool(((lm == 0xFFFF) == (ent < 0)).all())
    assert bool((lm[ent >= 0] < counts.long()[:, None, None].expand_as(lm)[ent >= 0]).all())
    named = torch.gather(hl.halo_rows.long(), 1, lm.clamp(max=hl.capacity - 1).view(T, -1)).view_as(lm)
    assert bool((named[ent >= 0] == ent[ent >= 0]).all())
    for t in (0, T // 2, T - 1):
        h = hl.halo_rows[t, :int(counts[t])]
        assert h.unique().numel() == h.numel() == ent[t][ent[t] >= 0].unique().numel()
    plain, _ = run(0)
    assert calls['halo'] == 2
    for name, a, b in zip(('out', 'BatchNorm sums', 'grad feats', 'grad conv1', 'grad gamma', 'grad beta', 'grad conv2'), got, plain):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7, name


@pytest.mark.parametrize('case', [
    dict(kernel=(3, 3, 3), cin=32, cout=128, n_cells=0.5, shape=(10, 24, 24), B=2),      # one 32-channel chunk
    dict(kernel=(3, 3, 3), cin=96, cout=64, n_cells=0.3, shape=(12, 20, 28), B=3),       # three chunks, 64 columns
    dict(kernel=(1, 3, 3), cin=64, cout=128, n_cells=0.6, shape=(4, 40, 40), B=2),       # nine offsets (a 2D neighbourhood)
    dict(kernel=(3, 3, 3), cin=128, cout=128, n_cells=0.02, shape=(16, 48, 48), B=1),    # almost no neighbours: mostly empty offsets
    dict(kernel=(3, 3, 3), cin=64, cout=64, n_cells=0.9, shape=(3, 7, 9), B=1),          # fewer rows than one tile
])
@pytest.mark.parametrize('flip', [0, 1])
def test_halo_entry_point_against_the_default_kernel(case, flip):
    """gga_sparse_halo_build + gga_sparse_conv_apply_halo called directly (C ABI) against gga_sparse_conv_apply_stats on the same
    rule book, weights and absmax slots: output rows and the BatchNorm sums, forward (flip 0) and backward-data (flip 1) order of
    the offsets - channel chunk counts 1 / 2 / 3 / 4, 9 and 27 offsets, dense and nearly empty levels, a level smaller than a tile."""
    from gga_amd import _lib, dense_conv, functional as F
    from gga_amd.sparse import _Level, _pack_weight, _Halo
    torch.manual_seed(3)
    dev, L = DEV, _lib.lib()
    B, shape = case['B'], case['shape']
    coors = (torch.rand(B, *shape) < case['n_cells']).nonzero().int()
    coors = coors[torch.randperm(len(coors))].contiguous().to(dev)
    n, cin, cout = coors.shape[0], case['cin'], case['cout']
    kvol = case['kernel'][0] * case['kernel'][1] * case['kernel'][2]
    lvl = _Level(coors, shape, B)
    rb = lvl.subm_rulebook(case['kernel'])
    halo = _Halo(coors, rb)
    assert halo.n_tiles == (n + 255) // 256 and int(halo.counts.min()) >= 1
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(kvol, cin, cout, device=dev) * 0.1
    x_amax, w_amax = dense_conv._amax_bits(x), dense_conv._amax_bits(w)
    wp = _pack_weight(w, kvol, cin, cout, 0, w_amax=w_amax)
    tiles = int(L.gga_sparse_conv_apply_tiles(n))
    y0, y1 = torch.empty(n, cout, device=dev), torch.full((n, cout), float('nan'), device=dev)
    s0 = torch.empty((tiles, 2, cout), dtype=torch.float64, device=dev)
    s1 = torch.full_like(s0, float('nan'))
    _lib.check(L.gga_sparse_conv_apply_stats(F._p(x), F._p(rb.nbr), F._p(wp), F._p(rb.perm), F._p(rb.mask), n, kvol, cin, cout, flip,
                                             F._p(y0), cout, 2, F._p(x_amax), F._p(w_amax), F._p(s0), F._stream()), 'default kernel')
    _lib.check(L.gga_sparse_conv_apply_halo(F._p(x), F._p(wp), F._p(halo.tile_rows), F._p(halo.counts), halo.capacity, F._p(halo.halo_rows),
                                            F._p(halo.local_map), n, halo.n_tiles, kvol, cin, cout, flip, F._p(y1), cout, 2, F._p(x_amax),
                                            F._p(w_amax), F._p(s1), None, 0, None, None, None, None, F._stream()), 'halo form')
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y1).all()) and bool(torch.isfinite(s1).all()), 'rows or statistics the halo form did not write'
    assert float((y0 - y1).abs().max()) <= 1e-5 * float(y0.abs().max()) + 1e-9
    a, b = s0.sum(0), s1.sum(0)
    assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-9


def test_sparse_map_straight_into_channels_last_memory():
    """functional.sparse_bev_channels_last (gga_sparse_bev_nhwc_fwd / _bwd) against SparseConvTensor.dense().view(N, C * D, H, W):
    values, memory format and the gradient of the features - and SparseEncoder(channels_last=True) through it equals the
    NCHW-scatter-then-copy path of rounds 1-2 bit for bit, forward and backward."""
    import copy
    from gga_amd import functional as F, sparse, sparse_encoder
    torch.manual_seed(2)
    B, shape, C = 3, (2, 26, 20), 128
    coors = (torch.rand(B, *shape) < 0.3).nonzero().int()
    coors = coors[torch.randperm(len(coors))].contiguous().to(DEV)
    f = torch.randn(len(coors), C, device=DEV, requires_grad=True)
    f2 = f.detach().clone().requires_grad_(True)
    ref = sparse.SparseConvTensor(f2, coors, shape, B).dense().view(B, C * shape[0], shape[1], shape[2])
    got = F.sparse_bev_channels_last(f, coors, B, *shape)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, ref)
    g = torch.randn_like(ref)
    got.backward(g.contiguous(memory_format=torch.channels_last))
    ref.backward(g)
    assert torch.equal(f.grad, f2.grad)
    # the encoder, both ways
    enc = SparseEncoder(in_channels=4, sparse_shape=[41, 64, 64], order=('conv', 'norm', 'act'), channels_last=True).to(DEV)
    c0 = torch.unique(torch.stack([torch.randint(0, 2, (3000,)), torch.randint(0, 41, (3000,)), torch.randint(0, 64, (3000,)),
                                   torch.randint(0, 64, (3000,))], 1), dim=0).int().to(DEV)
    v = torch.randn(len(c0), 4, device=DEV)
    outs = []
    for direct in (True, False):
        sparse_encoder.DIRECT_BEV = direct
        e = copy.deepcopy(enc)
        y = e(v, c0, 2)
        assert y.is_contiguous(memory_format=torch.channels_last)
        y.backward(torch.linspace(-1, 1, y.numel(), device=DEV).view(y.shape).contiguous(memory_format=torch.channels_last))
        outs.append((y.detach(), [p.grad.clone() for p in e.parameters()]))
    sparse_encoder.DIRECT_BEV = True
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


def test_index_plan_dies_with_its_coordinates_without_the_cyclic_collector():
    """``SparseEncoder.build_indices`` hangs the plan (levels, rule books: 1.6 GB per bs-8 batch of the shipped config) on the
    coordinates it returns. It must die by reference counting when they do: hung on the very tensor its first level keeps it
    formed a cycle that only a generation-2 pass of the collector freed, every ~12 steps, and the allocator grew by a batch's
    worth of hipMalloc calls per step meanwhile (round 5, tools_dev/who_holds.py)."""
    import gc
    import weakref
    enc = SparseEncoder(in_channels=4, sparse_shape=[41, 160, 160], order=('conv', 'norm', 'act')).to(DEV)
    coors = _coords(2, (41, 160, 160), 3000, seed=4).to(DEV)
    coors.num_valid = torch.tensor([len(coors)], dtype=torch.int32, device=DEV)
    gc.collect()
    gc.disable()
    try:
        out = enc.build_indices(coors, 2)
        assert out.data_ptr() == coors.data_ptr() and out.num_valid is coors.num_valid and out.index_plan.level0.n == len(coors)
        y = enc(torch.rand(len(coors), 4, device=DEV), out, 2)            # forward picks the plan up
        assert y.shape[0] == 2
        plan, level = weakref.ref(out.index_plan), weakref.ref(out.index_plan.level0)
        del out, y
        assert plan() is None and level() is None
    finally:
        gc.enable()


@pytest.mark.gpu
@pytest.mark.parametrize('kvol,n', [(27, 1), (27, 1000), (27, 513_777), (32, 70_001), (8, 4097), (9, 250_000)])
def test_mask_order_is_the_stable_sort_of_the_masks(kvol, n):
    """gga_sparse_mask_order (an LSD radix sort over the kvol mask bits) against torch.sort(mask, stable=True): a stable sort has
    one answer - the same int32 order, ties in row order, the sign bit of a 32-offset mask included."""
    from gga_amd.sparse import mask_order
    g = torch.Generator().manual_seed(7 * kvol + n)
    few = torch.randint(0, 2 ** 31 - 1, (37,), generator=g, dtype=torch.int64)               # few distinct patterns: many ties
    m = few[torch.randint(0, 37, (n,), generator=g)]
    m[::3] = torch.randint(0, 2 ** 31 - 1, (len(m[::3]),), generator=g, dtype=torch.int64)
    m = m & ((1 << kvol) - 1)
    if kvol == 32:
        m = m | (torch.randint(0, 2, (n,), generator=g, dtype=torch.int64) << 31)
    mask = torch.from_numpy(m.numpy().astype(np.uint32).view(np.int32)).cuda()
    got = mask_order(mask, kvol)
    want = torch.sort(mask, stable=True)[1].int()
    assert got.dtype == torch.int32 and torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize('B,shape,n', [(1, (5, 7, 3), 60), (8, (11, 200, 176), 400_000), (3, (41, 1600, 1408), 250_000), (2, (1, 1, 9), 9)])
def test_morton_order_entry_point_equals_the_tensor_expression(B, shape, n):
    """gga_sparse_morton_order (key kernel + radix sort over the bits the level's extent can set) against the framework expression
    it replaces (64-bit keys, torch.argsort): distinct coordinates have one ascending order."""
    from gga_amd.sparse import morton_order
    D, H, W = shape
    g = torch.Generator().manual_seed(n + D)
    cells = B * D * H * W
    n = min(n, cells)
    flat = torch.randperm(cells, generator=g)[:n] if cells < 5_000_000 else torch.unique(torch.randint(0, cells, (2 * n,), generator=g))[:n]
    flat = flat[torch.randperm(flat.shape[0], generator=g)]
    coors = torch.stack([flat // (D * H * W), (flat // (H * W)) % D, (flat // W) % H, flat % W], 1).int().cuda()
    got = morton_order(coors, B, max(shape))
    want = morton_order(coors)                                   # no extent: the tensor expression
    assert got.dtype == torch.int32 and torch.equal(got.long(), want)
