"""Host-side logic of the drop-in boundary, on CPU: registry / config surface, module and
state_dict naming, target packing against the oracle + golden vectors, schedules."""
import os

import numpy as np
import pytest
import torch

import gga_amd
from conftest import REPO, TRAIN_CFG, load_head_case
from gga_amd import Config, build_model, synthetic
from gga_amd.registry import MODELS, Registry, build_from_cfg
from gga_amd.train import CyclicSchedule
from oracle import oracle as O

PP_CFG = os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py')


def test_registry_semantics():
    for name in ('GGA', 'MVXTwoStageDetector_GGA', 'CenterHead_GGA', 'SeparateHead', 'HardSimpleVFE',
                 'PillarFeatureNet', 'PointPillarsScatter', 'SECOND', 'SECONDFPN', 'GaussianFocalLoss', 'L1Loss'):
        assert MODELS.get(name) is not None, name
    r = Registry('t')

    @r.register_module()
    class A:
        def __init__(self, x, y=2):
            self.x, self.y = x, y
    a = r.build(dict(type='A', x=1), default_args=dict(y=5))
    assert (a.x, a.y) == (1, 5)
    with pytest.raises(KeyError, match='is not in the t registry'):
        r.build(dict(type='B'))
    with pytest.raises(KeyError, match='already registered'):
        r.register_module(module=A)
    with pytest.raises(KeyError, match='must contain the key "type"'):
        build_from_cfg(dict(x=1), r)


def test_config_base_and_delete():
    cfg = Config.fromfile(PP_CFG)
    assert cfg.model.type == 'GGA'
    assert cfg.model.pts_voxel_encoder.type == 'PillarFeatureNet' and 'num_features' not in cfg.model.pts_voxel_encoder
    assert cfg.model.pts_bbox_head.in_channels == 384 and cfg.model.pts_bbox_head.share_conv_channel == 64
    assert cfg.model.train_cfg.pts.grid_size == [432, 496, 1] and cfg.model.train_cfg.pts.max_objs == 500
    assert cfg.optimizer.type == 'AdamW' and cfg.optimizer_config.grad_clip.max_norm == 35
    cfg.merge_from_dict({'optimizer.lr': 0.1})
    assert cfg.optimizer.lr == 0.1 and cfg.optimizer.weight_decay == 0.01


@pytest.fixture(scope='module')
def pp_model():
    torch.manual_seed(0)
    return build_model(Config.fromfile(PP_CFG).model)


def test_pp_model_structure(pp_model):
    sd = pp_model.state_dict()
    for k in ('pts_voxel_encoder.pfn_layers.0.linear.weight', 'pts_backbone.blocks.2.15.weight',
              'pts_neck.deblocks.2.0.weight', 'pts_bbox_head.shared_conv.conv.weight',
              'pts_bbox_head.shared_conv.bn.running_var', 'pts_bbox_head.task_heads.2.heatmap.1.bias',
              'pts_bbox_head.task_heads.0.rot.0.conv.weight'):
        assert k in sd, k
    assert sd['pts_voxel_encoder.pfn_layers.0.linear.weight'].shape == (64, 10)
    assert sd['pts_neck.deblocks.2.0.weight'].shape == (256, 128, 4, 4)
    assert torch.all(sd['pts_bbox_head.task_heads.1.heatmap.1.bias'] == -2.19)
    assert pp_model.pts_bbox_head.train_cfg['out_size_factor'] == 2
    assert pp_model.pts_voxel_layer.max_voxels == (16000, 40000)


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference tree only exists in the build container')
def test_reference_config_model_section_loads_unchanged():
    cfg = Config.fromfile('/root/reference/configs/gga/gga_kitti_config.py')
    mine = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
    def norm(x):
        if isinstance(x, dict):
            return {k: norm(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [norm(v) for v in x]
        return x
    assert norm(cfg.model) == norm(mine.model)
    for k in ('optimizer', 'optimizer_config', 'lr_config', 'momentum_config', 'runner'):
        assert norm(cfg[k]) == norm(mine[k]), k
    # every non-sparse stage of the reference's own file builds through the registry
    m = dict(cfg.model)
    head = dict(m['pts_bbox_head'], train_cfg=cfg.model.train_cfg.pts, test_cfg=cfg.model.test_cfg.pts)
    assert type(gga_amd.registry.build_head(head)).__name__ == 'CenterHead_GGA'
    assert type(gga_amd.registry.build_backbone(m['pts_backbone'])).__name__ == 'SECOND'
    assert type(gga_amd.registry.build_neck(m['pts_neck'])).__name__ == 'SECONDFPN'
    assert type(gga_amd.registry.build_voxel_encoder(m['pts_voxel_encoder'])).__name__ == 'HardSimpleVFE'


@pytest.mark.parametrize('c', ['second', 'pp'])
def test_pack_targets_matches_oracle_and_golden(golden, pp_model, c):
    d = golden('head')
    case = load_head_case(d, c)
    head = pp_model.pts_bbox_head
    saved = head.train_cfg
    head.train_cfg = TRAIN_CFG[c]
    try:
        torch.manual_seed(1234)
        pk = head.pack_targets(case['labels'], case['boxes_img'], case['lidar2img'], case['pseudo'], case['bdry'],
                               case['ibp'], [dict(lidar2img=m) for m in case['meta_l2i']])
    finally:
        head.train_cfg = saved
    torch.manual_seed(1234)
    srl = O.draw_srl(case['B'])
    assert np.array_equal(pk['srl'], srl)          # same CPU-generator draws, same order (head:514-525)
    tg = O.get_targets(case['labels'], case['boxes_img'], case['lidar2img'], case['pseudo'], case['bdry'],
                       case['ibp'], case['meta_l2i'], TRAIN_CFG[c], srl)
    B, K = case['B'], 500
    for t in range(3):
        assert np.array_equal(pk['ind'][t], d[f'{c}.tgt.{t}.ind'])
        assert np.array_equal(pk['mask'][t], d[f'{c}.tgt.{t}.mask'])
        assert np.array_equal(pk['bound_mask'][t], d[f'{c}.tgt.{t}.bound_mask'])
        assert np.array_equal(pk['lidar2img'][t], d[f'{c}.tgt.{t}.lidar2img'])
        np.testing.assert_array_equal(pk['anno_box'][t], d[f'{c}.tgt.{t}.anno_box'])
    # the splat list reproduces the reference heat maps when drawn by the oracle
    hm = np.zeros((3 * B, pk['fh'], pk['fw']), np.float32)
    for m, cx, cy, r in pk['objs']:
        O.draw_gaussian(hm[m], cx, cy, r)
    for t in range(3):
        np.testing.assert_array_equal(hm[t * B:(t + 1) * B, None], tg['heatmap'][t])
    # packed in-box points: task-major, slot = b*K + k, xy in f32
    o = 0
    for t in range(3):
        for b in range(B):
            for k, p in enumerate(tg['ibp'][t][b]):
                s, e = pk['ibp_offsets'][o], pk['ibp_offsets'][o + 1]
                assert pk['ibp_slot'][o] == b * K + k
                np.testing.assert_array_equal(pk['ibp_xy'][s:e], np.asarray(p)[:, :2].astype(np.float32))
                o += 1
        assert o == pk['task_nobj'][:t + 1].sum()


def test_cyclic_schedule():
    # lr: x10 up over 40 % of the run, then down to 1e-4 x base (gga_kitti_config.py:237-247)
    s = CyclicSchedule(1.5e-3, 1000, (10, 1e-4), 1, 0.4)
    assert s(0) == pytest.approx(1.5e-3)
    assert s(400) == pytest.approx(1.5e-2)
    assert s(200) == pytest.approx((1.5e-3 + 1.5e-2) / 2)
    assert 1.5e-7 < s(999) < 3e-7              # one step before the 1e-4 floor
    assert all(s(i) <= s(i + 1) for i in range(0, 399)) and all(s(i) >= s(i + 1) for i in range(400, 998))
    m = CyclicSchedule(0.95, 1000, (0.85 / 0.95, 1), 1, 0.4)
    assert m(400) == pytest.approx(0.85) and m(0) == pytest.approx(0.95)


def test_schedules_match_the_oracle_restatement():
    """The product's schedules (gga_amd/train.py) against the oracle's independent restatement of mmcv's cyclic and
    step hooks (oracle/torch_ref.py) at every iteration of a run."""
    from gga_amd.train import StepSchedule
    from oracle import torch_ref as R
    for base, n, ratio, times, up in ((1.5e-3, 1000, (10, 1e-4), 1, 0.4), (0.95, 1000, (0.85 / 0.95, 1), 1, 0.4),
                                      (1e-3, 990, (8, 1e-3), 3, 0.3), (0.9, 77, (0.8, 1.0), 1, 0.5)):
        s = CyclicSchedule(base, n, ratio, times, up)
        for it in range(n):
            assert s(it) == pytest.approx(R.cyclic_value(base, it, n, ratio, times, up), rel=1e-12, abs=0), (base, it)
    st = StepSchedule(1e-3, [32, 44], iters_per_epoch=25, gamma=0.1, warmup='linear', warmup_iters=500, warmup_ratio=1 / 3)
    for it in range(0, 48 * 25):
        assert st(it) == pytest.approx(R.step_value(1e-3, it, [32 * 25, 44 * 25], 0.1, 500, 1 / 3), rel=1e-12), it


def test_synthetic_frame_contract():
    f = synthetic.make_frame(3)
    g = synthetic.make_frame(3)
    assert torch.equal(f['points'], g['points'])                     # seeded
    assert f['points'].shape == (20000, 4) and f['points'].dtype == torch.float32
    n = len(f['gt_labels_3d'])
    assert 4 <= n <= 20 and f['GGA_lidar2img'].shape == (n, 4, 4) and f['GGA_init_pseudo_labels'].dtype == torch.float64
    assert f['GGA_bdry_masks'].dtype == torch.bool and len(f['GGA_in_box_points']) == n
    x, y, z = f['points'][:, 0], f['points'][:, 1], f['points'][:, 2]
    outside = (x < 0) | (x >= 70.4) | (y < -40) | (y >= 40) | (z < -3) | (z >= 1)
    assert 0.03 < outside.float().mean() < 0.07                      # the rejection branch is exercised


def test_sparse_conv_loads_spconv2_checkpoint_layout():
    """write_spconv2.py:42-101: spconv 2.x keeps kernels as [Cout, kz, ky, kx, Cin] (state_dict version 2),
    mmcv as [kz, ky, kx, Cin, Cout]; both load into the sparse conv layers here."""
    from collections import OrderedDict
    from gga_amd.sparse import SparseConv3d, SubMConv3d
    torch.manual_seed(0)
    src = SubMConv3d(16, 32, 3, bias=False)
    # mmcv layout: plain load
    dst = SubMConv3d(16, 32, 3, bias=False)
    dst.load_state_dict(src.state_dict())
    assert torch.equal(dst.weight, src.weight)
    # spconv 2 layout with its version stamp
    sd = OrderedDict(weight=src.spconv2_weight())
    assert tuple(sd['weight'].shape) == (32, 3, 3, 3, 16)
    sd._metadata = OrderedDict({'': dict(version=2)})
    dst2 = SubMConv3d(16, 32, 3, bias=False)
    dst2.load_state_dict(sd)
    assert torch.equal(dst2.weight, src.weight)
    # ... and without the stamp (recognised by shape), anisotropic kernel, inside a parent module
    conv = SparseConv3d(8, 8, (3, 1, 1), stride=(2, 1, 1), bias=False)
    parent = torch.nn.Sequential(OrderedDict(down=conv))
    sd = {'down.weight': conv.spconv2_weight()}
    twin = torch.nn.Sequential(OrderedDict(down=SparseConv3d(8, 8, (3, 1, 1), stride=(2, 1, 1), bias=False)))
    twin.load_state_dict(sd)
    assert torch.equal(twin.down.weight, conv.weight)
    # a genuinely wrong shape still fails loudly
    with pytest.raises(RuntimeError, match='size mismatch'):
        SubMConv3d(16, 32, 3, bias=False).load_state_dict({'weight': torch.zeros(3, 3, 3, 16, 8)})


def test_pgd_config_equals_reference_and_builds():
    """configs/gga/gga_pdg.py (the `_base_` chain merged by hand) resolves to the same model / optimizer /
    schedule sections as the reference's file, and builds through the registries: FCOSMono3D = ResNet-101
    (caffe style) + FPN + PGDHead with DCNv2 on the last tower convolutions."""
    ref_path = '/root/reference/configs/gga/gga_pdg.py'
    mine = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_pdg.py'))
    if os.path.exists(ref_path):
        ref = Config.fromfile(ref_path)
        plain = lambda v: {k: plain(x) for k, x in v.items()} if isinstance(v, dict) else \
            [plain(x) for x in v] if isinstance(v, (list, tuple)) else v
        for key in ('model', 'optimizer', 'optimizer_config', 'lr_config', 'total_epochs', 'runner', 'evaluation',
                    'checkpoint_config'):
            assert plain(mine[key]) == plain(ref[key]), key
        assert mine.data['samples_per_gpu'] == ref.data['samples_per_gpu'] == 12
        assert plain(mine.data['train']['ann_file']) == plain(ref.data['train']['ann_file'])
    model = build_model(mine.model)
    from gga_amd.dcn import ModulatedDeformConv2dPack
    from gga_amd.mono3d_detectors import FCOSMono3D
    assert isinstance(model, FCOSMono3D)
    assert isinstance(model.bbox_head.cls_convs[-1].conv, ModulatedDeformConv2dPack)
    names = set(dict(model.named_parameters()))
    for n in ('backbone.layer3.22.conv3.weight', 'backbone.layer1.0.downsample.0.weight', 'neck.lateral_convs.3.conv.weight',
              'neck.fpn_convs.0.conv.bias', 'bbox_head.cls_convs.1.conv.conv_offset.weight', 'bbox_head.conv_regs.4.weight',
              'bbox_head.scales.3.4.scale', 'bbox_head.fuse_lambda', 'bbox_head.conv_weights.0.bias'):
        assert n in names, n
    # frozen BatchNorm statistics (norm_eval) and frozen stem (frozen_stages=0)
    model.train()
    assert not model.backbone.bn1.training and not model.backbone.conv1.weight.requires_grad
    assert not model.backbone.layer2[0].bn2.weight.requires_grad and model.backbone.layer2[0].conv2.weight.requires_grad
    from gga_amd.train import StepSchedule, build_optimizer
    opt = build_optimizer(model, mine.optimizer)
    by_lr = {round(g['lr'], 6) for g in opt.param_groups}
    assert by_lr == {0.001, 0.002} and all(g['weight_decay'] == 0 for g in opt.param_groups if g['lr'] == 0.002)
    s = StepSchedule(1.0, [32, 44], iters_per_epoch=100, warmup='linear', warmup_iters=500, warmup_ratio=1 / 3)
    assert s(0) == pytest.approx(1 / 3) and s(250) == pytest.approx(2 / 3) and s(500) == 1.0
    assert s(3199) == 1.0 and s(3200) == pytest.approx(0.1) and s(4400) == pytest.approx(0.01)


def test_setup_multi_processes_like_the_reference(monkeypatch):
    """mmdet3d/utils/setup_env.py:10-53 (tests/test_utils/test_setup_env.py): with more than one data-loader worker per GPU
    OMP_NUM_THREADS / MKL_NUM_THREADS default to 1 per process, an existing setting is kept, one worker changes nothing."""
    import torch
    from gga_amd import Config
    from gga_amd.train import setup_multi_processes
    before = torch.get_num_threads()
    try:
        monkeypatch.delenv('OMP_NUM_THREADS', raising=False)
        monkeypatch.delenv('MKL_NUM_THREADS', raising=False)
        setup_multi_processes(Config(dict(data=dict(workers_per_gpu=1))))
        assert 'OMP_NUM_THREADS' not in os.environ and 'MKL_NUM_THREADS' not in os.environ
        monkeypatch.setenv('OMP_NUM_THREADS', '4')
        setup_multi_processes(Config(dict(data=dict(workers_per_gpu=2))))
        assert os.environ['OMP_NUM_THREADS'] == '4' and os.environ['MKL_NUM_THREADS'] == '1'
        monkeypatch.delenv('OMP_NUM_THREADS')
        monkeypatch.delenv('MKL_NUM_THREADS')
        setup_multi_processes(Config(dict(data=dict(workers_per_gpu=1, train_dataloader=dict(workers_per_gpu=4)))))
        assert os.environ['OMP_NUM_THREADS'] == '1' and os.environ['MKL_NUM_THREADS'] == '1' and torch.get_num_threads() == 1
    finally:
        limit = getattr(setup_multi_processes, '_blas_limit', None)
        if limit is not None:
            limit.restore_original_limits()
            setup_multi_processes._blas_limit = None
        torch.set_num_threads(before)


def test_pgd_optimizer_groups_follow_mmcv_paramwise_rules():
    """configs/gga/gga_pdg.py: SGD lr 1e-3, wd 1e-4, paramwise_cfg(bias_lr_mult=2, bias_decay_mult=0). mmcv's
    DefaultOptimizerConstructor (un-vendored, restated): the multipliers reach every ``bias`` EXCEPT those of normalisation
    layers (GroupNorm of the towers: base lr, weight decay kept) and of a DCN module's conv_offset; the DCN module's own bias
    takes them."""
    from gga_amd.train import build_optimizer, paramwise_settings
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_pdg.py'))
    model = build_model(cfg.model)
    st = paramwise_settings(model, 1e-3, 1e-4, cfg.optimizer['paramwise_cfg'])
    names = dict(model.named_parameters())
    assert set(st) == {n for n, p in names.items() if p.requires_grad}
    gn_bias = [n for n in st if '.gn.bias' in n]
    gn_weight = [n for n in st if '.gn.weight' in n]
    assert len(gn_bias) > 10 and all(st[n] == (1e-3, 1e-4) for n in gn_bias + gn_weight)
    off = [n for n in st if 'conv_offset' in n]
    assert len(off) == 4 and all(st[n] == (1e-3, 1e-4) for n in off)             # weight and bias of both towers' DCN offsets
    dcn_bias = [n[:-len('conv_offset.bias')] + 'bias' for n in off if n.endswith('conv_offset.bias')]
    assert all(st[n] == (2e-3, 0.0) for n in dcn_bias)
    conv_bias = [n for n in st if n.endswith('.bias') and n not in gn_bias and 'conv_offset' not in n]
    assert len(conv_bias) > 20 and all(st[n] == (2e-3, 0.0) for n in conv_bias)
    assert all(st[n] == (1e-3, 1e-4) for n in st if n.endswith('.weight') or n.endswith('scale'))
    # the frozen backbone norms (requires_grad=False) are in no group
    opt = build_optimizer(model, cfg.optimizer)
    in_opt = {id(p) for g in opt.param_groups for p in g['params']}
    assert in_opt == {id(p) for p in names.values() if p.requires_grad}
    assert sorted((g['lr'], g['weight_decay']) for g in opt.param_groups) == [(1e-3, 1e-4), (2e-3, 0.0)]


def test_resnet_honours_init_cfg_like_mmdet(tmp_path):
    """mmdet's ResNet zeroes the last norm of each block only when there is NO init_cfg; with a Pretrained init_cfg
    (configs/gga/gga_pdg.py:11) nothing is zeroed and init_weights loads the checkpoint - from a local file here; a model-zoo
    URL that cannot be fetched is reported, not silently dropped."""
    import warnings
    import torch
    from gga_amd.registry import build_backbone
    base = dict(type='ResNet', depth=50, num_stages=2, strides=(1, 2), out_indices=(0, 1), frozen_stages=1,
                norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='caffe')
    plain = build_backbone(dict(base))
    assert all(float(b.bn3.weight.abs().max()) == 0 for b in plain.modules() if hasattr(b, 'bn3'))
    donor = build_backbone(dict(base, zero_init_residual=False))
    with torch.no_grad():
        donor.layer2[1].conv2.weight.fill_(0.125)
    path = tmp_path / 'resnet.pth'
    torch.save({'state_dict': donor.state_dict()}, path)
    net = build_backbone(dict(base, init_cfg=dict(type='Pretrained', checkpoint=str(path))))
    assert all(float(b.bn3.weight.min()) == 1 for b in net.modules() if hasattr(b, 'bn3'))       # not zeroed
    net.init_weights()
    assert float(net.layer2[1].conv2.weight.min()) == 0.125
    zoo = build_backbone(dict(base, init_cfg=dict(type='Pretrained', checkpoint='open-mmlab://detectron2/resnet101_caffe')))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        zoo.init_weights()
    assert any('cannot be downloaded' in str(x.message) for x in w)
    assert all(float(b.bn3.weight.min()) == 1 for b in zoo.modules() if hasattr(b, 'bn3'))


def test_runner_selects_two_planes_and_prefetch_cache_is_keyed_by_identity():
    """Runner: the config / default selects the two-fp16-plane arithmetic unless GGA_DENSE_PLANES pins it; a prefetched batch
    is only ever used for the very dict (and points list) it was made from."""
    from gga_amd import dense_conv
    from gga_amd.train import Runner
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
    model = build_model(cfg.model)
    dense_conv.PLANES, dense_conv.PLANES_PINNED, dense_conv.FELL_BACK = 3, False, False
    r = Runner(model, cfg, max_iters=10)
    assert r.planes == 2 and dense_conv.PLANES == 3 and r.range_check_interval == 500       # the Runner's own choice: the process keeps its default
    dense_conv.PLANES, dense_conv.PLANES_PINNED = 3, True
    assert Runner(model, cfg, max_iters=10).planes == 3                                      # GGA_DENSE_PLANES pins it for everybody
    cfg['gga_dense_planes'] = 3
    dense_conv.PLANES, dense_conv.PLANES_PINNED = 2, False
    assert Runner(model, cfg, max_iters=10).planes == 3
    cfg['gga_dense_planes'] = 2
    dense_conv.FELL_BACK = True                                                              # a guard fell back earlier in this process
    assert Runner(model, cfg, max_iters=10).planes == 3
    dense_conv.FELL_BACK = False
    a = dict(points=[1, 2])
    r._prepared[id(a)] = ('prep', 'ev', a, a['points'])
    assert r._prepared_for(a)[0] == 'prep'
    b = dict(a)                       # another dict (even with the same list) is not the prefetched batch
    assert r._prepared_for(b) is None
    a['points'] = [3]                 # the same dict with other points is not either
    assert r._prepared_for(a) is None


def test_runner_inputs_ready_is_a_no_op_without_a_gpu():
    """Runner.inputs_ready announces resident batches with a stream event; on a CPU device there is no stream and nothing to
    announce (and nothing recorded that a later prefetch would wait for)."""
    from gga_amd.train import Runner
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
    r = Runner(build_model(cfg.model), cfg, max_iters=10)
    r.inputs_ready(dict(points=[1]), dict(points=[2]))
    assert r._ready == {}


def test_batchnorm_counters_deferred_into_one_add():
    """functional.deferred_batch_counters: the counters a forward pass would increment are collected and added once at
    the end (a layer applied twice counts twice); outside it ``count_batch`` increments at once (reference semantics:
    torch.nn.modules.batchnorm._BatchNorm.forward, ``num_batches_tracked += 1`` per call in training mode)."""
    import torch
    from gga_amd import functional as F
    a, b = torch.nn.BatchNorm1d(4), torch.nn.BatchNorm2d(4)
    with F.deferred_batch_counters():
        F.count_batch(a), F.count_batch(b), F.count_batch(a)
        assert int(a.num_batches_tracked) == 0 and int(b.num_batches_tracked) == 0
        with F.deferred_batch_counters():              # nested passes keep their own lists
            F.count_batch(b)
        assert int(b.num_batches_tracked) == 1
    assert int(a.num_batches_tracked) == 2 and int(b.num_batches_tracked) == 2
    F.count_batch(a)
    assert int(a.num_batches_tracked) == 3
    with F.deferred_batch_counters():                  # an empty pass adds nothing
        pass
    assert int(a.num_batches_tracked) == 3


def test_upload_many_hands_back_every_array_as_a_view_of_one_buffer():
    """functional.upload_many (the head's target arrays: one staging buffer, one copy): every array comes back with its dtype,
    shape and values - empty ones and odd sizes included - as views of ONE buffer at 256-byte offsets."""
    from gga_amd import functional as F
    g = np.random.default_rng(3)
    arrays = dict(a=g.standard_normal((3, 2, 5)).astype(np.float32), b=g.integers(0, 1 << 40, (7,)).astype(np.int64),
                  c=(g.random((2, 3, 4)) < 0.5).astype(np.uint8), d=np.zeros((0, 2), np.float32), e=np.arange(5, dtype=np.int32),
                  f=g.standard_normal((4, 4)).astype(np.float64)[:, ::2])            # not contiguous
    out = F.upload_many(arrays, 'cpu')
    base = {t.untyped_storage().data_ptr() for t in out.values()}
    assert len(base) == 1
    for k, a in arrays.items():
        t = out[k]
        assert tuple(t.shape) == a.shape and t.dtype == torch.from_numpy(np.empty(0, a.dtype)).dtype
        assert np.array_equal(t.numpy(), a)
        assert t.numel() == 0 or (t.data_ptr() - next(iter(base))) % 256 == 0


def test_seed_helpers_of_the_train_api():
    """``init_random_seed`` / ``set_random_seed`` (mmdet3d/apis/train.py:27-74): a given seed is returned as it is, none gives a
    fresh one; after ``set_random_seed`` python's, numpy's and torch's generators repeat - the three the train entry's objects draw
    from (weight initialisation, database sampler order, SRL coefficients)."""
    import random
    from gga_amd.train import init_random_seed, set_random_seed
    assert init_random_seed(7) == 7 and isinstance(init_random_seed(), int)

    def draws():
        return random.random(), float(np.random.rand()), float(torch.rand(())), float(torch.normal(torch.tensor(1.35), torch.tensor(0.48)))
    set_random_seed(3)
    a = draws()
    set_random_seed(3)
    assert draws() == a
    set_random_seed(4)
    assert draws() != a
