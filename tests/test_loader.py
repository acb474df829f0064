"""The dataset-fed train loop (mmdet3d/apis/train.py:205-219,275-285,318-322): dataset wrappers, samplers, collate, the
hand-over to ``Runner.step``, checkpoints and resume. The sampler / collate semantics are mmdet's / mmcv's (third-party,
restated: parity unpinned) - what is tested is what the path needs from them: every frame once per epoch over the ranks,
disjoint shards, a new order per epoch, the per-frame list layout ``forward_train`` takes, and that a resumed run continues
exactly where the saved one stopped."""
import copy
import os
import pickle
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO
from gga_amd import Config, synthetic
from gga_amd import loader as LD
from gga_amd.pipelines import DataContainer as DC
from gga_amd.train import Runner, find_latest_checkpoint, train_detector

SEEDS = (71, 72, 73)
CLASSES = ['Pedestrian', 'Cyclist', 'Car']
PP_RANGE = [0, -39.68, -3, 69.12, 39.68, 1]


def kitti_tree(root):
    """The on-disk fixture of tests/test_datasets.py: three synthetic KITTI frames + the info file the reference's
    converter produced for them."""
    os.makedirs(os.path.join(root, 'training', 'velodyne'), exist_ok=True)
    for seed in SEEDS:
        synthetic.make_rga_scene(seed)[0].tofile(os.path.join(root, 'training', 'velodyne', f'{seed:06d}.bin'))
    return pickle.load(open(os.path.join(GOLDEN, 'gt_database_infos.pkl'), 'rb'))


def train_pipeline(point_range=PP_RANGE, min_points=1):
    """configs/gga/gga_kitti_config.py's train_pipeline without the database sampler (its database needs the device)."""
    return [dict(type='LoadPointsFromFile', coord_type='LIDAR', load_dim=4, use_dim=4),
            dict(type='LoadAnnotations3D', with_bbox_3d=True, with_label_3d=True, with_bbox=True, with_gga=True),
            dict(type='PointsRangeFilter', point_cloud_range=point_range),
            dict(type='ObjectRangeFilter_GGA', point_cloud_range=point_range, num_points_range=min_points),
            dict(type='PointShuffle'),
            dict(type='DefaultFormatBundle3D_GGA', class_names=CLASSES),
            dict(type='Collect3D_GGA', keys=['points', 'gt_bboxes_3d', 'gt_labels_3d', 'GGA_boxes_img', 'GGA_lidar2img',
                                             'GGA_init_pseudo_labels', 'GGA_bdry_masks', 'GGA_in_box_points'])]


def dataset_cfg(root, infos, times=4, **kw):
    return dict(type='RepeatDataset', times=times,
                dataset=dict(type='KittiDataset_GGA_train', data_root=root, ann_file=infos, split='training', pts_prefix='velodyne',
                             pipeline=train_pipeline(**kw), modality=dict(use_lidar=True, use_camera=False), classes=CLASSES,
                             test_mode=False, box_type_3d='LiDAR'))


class _Frames:
    def __init__(self, n, flag=None):
        self.n = n
        if flag is not None:
            self.flag = np.asarray(flag, dtype=np.uint8)

    def __len__(self):
        return self.n


def test_repeat_dataset_and_builder(tmp_path):
    infos = kitti_tree(str(tmp_path))
    ds = LD.build_dataset(dataset_cfg(str(tmp_path), infos, times=3))
    assert isinstance(ds, LD.RepeatDataset) and len(ds) == 9 and len(ds.dataset) == 3
    assert ds.CLASSES == tuple(CLASSES)
    np.random.seed(0), torch.manual_seed(0)
    a, b = ds[1], ds[4]
    assert a['img_metas'].data['sample_idx'] == b['img_metas'].data['sample_idx'] == SEEDS[1]
    assert isinstance(a['points'], DC) and a['points'].data.shape[1] == 4
    assert a['gt_bboxes_3d'].cpu_only and len(a['gt_bboxes_3d'].data) == len(a['gt_labels_3d'].data)


def test_distributed_group_sampler_shards_the_frames():
    ds = _Frames(23)
    world, spg = 4, 2
    samplers = [LD.DistributedGroupSampler(ds, spg, world, r, seed=5) for r in range(world)]
    shards = [list(s) for s in samplers]
    assert all(len(x) == len(samplers[0]) == 6 for x in shards)                  # ceil(23 / 8) * 8 = 24 -> 6 per rank
    seen = np.concatenate(shards)
    assert set(seen.tolist()) == set(range(23))                                 # every frame, 24 - 23 = one repeat
    assert len(seen) - len(set(seen.tolist())) == 1
    # a new order every epoch, the same order for the same (seed, epoch) on every rank's copy
    again = [list(LD.DistributedGroupSampler(ds, spg, world, r, seed=5)) for r in range(world)]
    assert again == shards
    for s in samplers:
        s.set_epoch(1)
    assert [list(s) for s in samplers] != shards
    assert [list(s) for s in samplers] != [list(LD.DistributedGroupSampler(ds, spg, world, r, seed=9)) for r in range(world)]          # (seed + epoch is what seeds the order)
    # batches never mix groups
    flags = np.array([0] * 10 + [1] * 7)
    s = LD.DistributedGroupSampler(_Frames(17, flags), 2, 2, 0, seed=0)
    idx = np.array(list(s)).reshape(-1, 2)
    assert (flags[idx[:, 0]] == flags[idx[:, 1]]).all()


def test_plain_samplers():
    ds = _Frames(10)
    parts = [list(LD.DistributedSampler(ds, 3, r, shuffle=False)) for r in range(3)]
    assert parts == [[0, 3, 6, 9], [1, 4, 7, 0], [2, 5, 8, 1]]                   # strided, padded by wrapping around
    shuffled = [list(LD.DistributedSampler(ds, 2, r, shuffle=True, seed=3)) for r in range(2)]
    assert sorted(shuffled[0] + shuffled[1]) == list(range(10))
    np.random.seed(1)
    g = list(LD.GroupSampler(_Frames(7), samples_per_gpu=3))
    assert len(g) == 9 and set(g) == set(range(7))


def test_collate_rules_and_step_inputs():
    mk = lambda n, k: dict(
        points=DC(torch.full((n, 4), float(k))), gt_labels_3d=DC(torch.arange(k)), gt_bboxes_3d=DC(f'boxes{k}', cpu_only=True),
        img_metas=DC(dict(sample_idx=k), cpu_only=True), img=DC(torch.ones(3, 2 + k, 5), stack=True, padding_value=7),
        GGA_in_box_points=DC([torch.zeros(2, 4)] * k))
    batch = LD.collate([mk(5, 1), mk(6, 2), mk(7, 3), mk(8, 4)], samples_per_gpu=2)
    assert [len(c) for c in batch['points'].data] == [2, 2] and batch['points'].data[1][0].shape == (7, 4)
    assert batch['gt_bboxes_3d'].cpu_only and batch['gt_bboxes_3d'].data == [['boxes1', 'boxes2'], ['boxes3', 'boxes4']]
    assert batch['img'].stack and batch['img'].data[0].shape == (2, 3, 4, 5) and batch['img'].data[1].shape == (2, 3, 6, 5)
    assert float(batch['img'].data[0][0, 0, 3, 0]) == 7.0 and float(batch['img'].data[0][1, 0, 3, 0]) == 1.0
    step = LD.to_step_inputs(batch, chunk=1)
    assert [m['sample_idx'] for m in step['img_metas']] == [3, 4] and [len(l) for l in step['gt_labels_3d']] == [3, 4]
    assert len(step['GGA_in_box_points'][1]) == 4 and step['points'][0].shape == (7, 4)
    assert LD.collate([(1, 'a'), (2, 'b')])[0].tolist() == [1, 2]


def test_packed_hand_over_gives_the_same_lists_and_the_same_targets(tmp_path):
    """``collate(packed=True)`` -> ``to_step_inputs``: the per-frame lists ``forward_train`` takes are the lists of the plain
    collate (values, dtypes, nesting; boxes as LiDARInstance3DBoxes), travelling as one tensor per key; and
    ``CenterHead_GGA.pack_targets`` returns identical arrays from either (its packed fast path against its list path)."""
    import pickle as pkl
    from gga_amd import build_model
    from gga_amd.box3d import LiDARInstance3DBoxes
    root = str(tmp_path)
    info_path, db_path = synthetic.write_kitti_tree(root, 6, n_points=3000, n_ibp_range=(20, 120), db_per_class=12)
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
    d = cfg.data['train']
    d['dataset'].update(data_root=root + '/', ann_file=info_path)
    for t in d['dataset']['pipeline']:
        if t['type'] == 'ObjectSample_GGA':
            t['db_sampler'].update(data_root=root + '/', info_path=db_path)
        if 'point_cloud_range' in t:
            t['point_cloud_range'] = PP_RANGE
    ds = LD.build_dataset(d)
    np.random.seed(3), torch.manual_seed(3)
    samples = [ds[i] for i in range(6)]
    assert max(len(s['gt_labels_3d'].data) for s in samples) > 12             # database objects were pasted
    plain = LD.to_step_inputs(LD.collate(samples, samples_per_gpu=3), chunk=1)
    packed = LD.collate(samples, samples_per_gpu=3, packed=True)
    wire = pkl.loads(pkl.dumps(packed))                                        # what crosses the process boundary
    n_tensors = lambda o: (1 if torch.is_tensor(o) else sum(n_tensors(x) for x in (o.values() if isinstance(o, dict) else o))
                           if isinstance(o, (list, tuple, dict)) else n_tensors([o.flat]) if isinstance(o, LD.PackedFrames)
                           else n_tensors(o.data) if isinstance(o, DC) else n_tensors([o.tensor]) if isinstance(o, LiDARInstance3DBoxes) else 0)
    assert n_tensors(wire) == 2 * 8 and n_tensors(LD.collate(samples, samples_per_gpu=3)) > 100
    # ... and all 16 payloads are views of ONE byte buffer: a single shared-memory segment per batch
    arenas = {p.arena.data_ptr() for v in wire.values() if isinstance(v, DC) for p in v.data if isinstance(p, LD.PackedFrames)}
    assert len(arenas) == 1
    import io
    from multiprocessing.reduction import ForkingPickler
    import torch.multiprocessing as _tmp  # noqa: F401  (registers the tensor reductions the loader's queue uses)
    buf = io.BytesIO()
    ForkingPickler(buf).dump(packed)
    assert buf.getvalue().count(b'rebuild_storage') == 1
    # the collate contract without to_step_inputs: a packed payload iterates / indexes as the per-frame list it stands for
    raw_pts, plain_pts = wire['points'].data[1], LD.collate(samples, samples_per_gpu=3)['points'].data[1]
    assert isinstance(raw_pts, LD.PackedFrames) and len(raw_pts) == 3
    assert all(torch.equal(a, b) for a, b in zip(raw_pts, plain_pts)) and torch.equal(raw_pts[2], plain_pts[2])
    got = LD.to_step_inputs(wire, chunk=1)
    assert set(got) == set(plain)
    for k, a in plain.items():
        b = got[k]
        if k == 'img_metas':
            assert [m['sample_idx'] for m in a] == [m['sample_idx'] for m in b]
            continue
        assert isinstance(b, LD.FrameList) and len(a) == len(b) == 3, k
        for fa, fb in zip(a, b):
            if k == 'gt_bboxes_3d':
                assert isinstance(fb, LiDARInstance3DBoxes) and torch.equal(fa.tensor, fb.tensor) and torch.equal(fa.gravity_center, fb.gravity_center)
            elif k == 'GGA_in_box_points':
                assert len(fa) == len(fb) and all(x.dtype == y.dtype and torch.equal(x, y) for x, y in zip(fa, fb))
            else:
                assert fa.dtype == fb.dtype and torch.equal(fa, fb), k
    head = build_model(cfg.model).pts_bbox_head
    srl = head.draw_srl(3)
    args = lambda x: (x['gt_labels_3d'], x['GGA_boxes_img'], x['GGA_lidar2img'], x['GGA_init_pseudo_labels'], x['GGA_bdry_masks'],
                      x['GGA_in_box_points'], x['img_metas'])
    ta, tb = head.pack_targets(*args(plain), srl=srl), head.pack_targets(*args(got), srl=srl)
    assert set(ta) == set(tb) and len(ta['ibp_xy']) > 500
    for k in ta:
        assert np.array_equal(np.asarray(ta[k]), np.asarray(tb[k])), k


def test_loader_from_the_kitti_tree(tmp_path):
    infos = kitti_tree(str(tmp_path))
    ds = LD.build_dataset(dataset_cfg(str(tmp_path), infos, times=2))
    np.random.seed(0), torch.manual_seed(0)
    loader = LD.build_dataloader(ds, samples_per_gpu=3, workers_per_gpu=0, dist=False, seed=0)
    batches = [LD.to_step_inputs(b) for b in loader]
    assert len(batches) == 2
    for b in batches:
        assert set(b) == {'img_metas', 'points', 'gt_bboxes_3d', 'gt_labels_3d', 'GGA_boxes_img', 'GGA_lidar2img',
                          'GGA_init_pseudo_labels', 'GGA_bdry_masks', 'GGA_in_box_points'}
        assert len(b['points']) == 3 and all(p.dtype == torch.float32 and p.shape[1] == 4 for p in b['points'])
        for boxes, labels, img, l2i, ibp in zip(b['gt_bboxes_3d'], b['gt_labels_3d'], b['GGA_boxes_img'], b['GGA_lidar2img'], b['GGA_in_box_points']):
            assert len(boxes) == len(labels) == len(img) == len(l2i) == len(ibp)
    assert sorted(m['sample_idx'] for b in batches for m in b['img_metas']) == sorted(SEEDS * 2)
    # worker processes: seeded per worker, the same frames come out
    loader = LD.build_dataloader(ds, samples_per_gpu=3, workers_per_gpu=2, dist=False, seed=0)
    assert sorted(m['sample_idx'] for b in loader for m in LD.to_step_inputs(b)['img_metas']) == sorted(SEEDS * 2)


# ------------------------------------------------------------------------------------------------ two ranks over gloo
CFG = dict(optimizer=dict(type='AdamW', lr=1e-2, betas=(0.95, 0.99), weight_decay=0.01),
           optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)),
           lr_config=dict(policy='cyclic', target_ratio=(10, 1e-4), cyclic_times=1, step_ratio_up=0.4),
           momentum_config=dict(policy='cyclic', target_ratio=(0.85 / 0.95, 1), cyclic_times=1, step_ratio_up=0.4),
           runner=dict(type='EpochBasedRunner', max_epochs=2), checkpoint_config=dict(interval=1), seed=11,
           data=dict(samples_per_gpu=2, workers_per_gpu=0), workflow=[('train', 1)])


def _tiny():
    from test_distributed_cpu import TinyDet
    return TinyDet()


class _Recording:
    """Dataset wrapper that writes down which frames this process loaded."""

    def __init__(self, ds, log):
        self.ds, self.log = ds, log
        self.flag = LD.group_flags(ds)

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, i):
        self.log.append(int(i))
        return self.ds[i]


def _rank_main(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    infos = pickle.load(open(os.path.join(GOLDEN, 'gt_database_infos.pkl'), 'rb'))
    seen = []
    ds = _Recording(LD.build_dataset(dataset_cfg(root, infos, times=4)), seen)
    cfg = Config(dict(CFG, work_dir=os.path.join(root, 'work')))
    model = _tiny()
    runner = train_detector(model, ds, cfg, distributed=True, device=torch.device('cpu'))
    q.put((rank, seen, runner.iter, runner.epoch, [p.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_train_from_the_dataset_with_disjoint_shards(tmp_path):
    kitti_tree(str(tmp_path))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, seen0, it0, ep0, w0), (_, seen1, it1, ep1, w1) = res
    assert it0 == it1 == 6 and ep0 == ep1 == 2                      # 12 items / (2 ranks x 2 per batch) = 3 iterations x 2 epochs
    # per epoch the two ranks load disjoint halves of the 12 items
    for e in range(2):
        a, b = set(seen0[6 * e:6 * e + 6]), set(seen1[6 * e:6 * e + 6])
        assert len(a) == len(b) == 6 and not (a & b) and (a | b) == set(range(12))
    assert seen0[:6] != seen0[6:]                                   # DistSamplerSeedHook: another order in epoch 2
    for a, b in zip(w0, w1):
        assert (a == b).all()                                       # all-reduced: the ranks hold the same parameters
    work = os.path.join(str(tmp_path), 'work')
    assert sorted(os.listdir(work)) == ['epoch_1.pth', 'epoch_2.pth', 'latest.pth']      # rank 0 only, every epoch
    assert os.path.realpath(find_latest_checkpoint(work)) == os.path.realpath(os.path.join(work, 'epoch_2.pth'))
    ck = torch.load(os.path.join(work, 'epoch_2.pth'), weights_only=False)
    assert ck['meta']['epoch'] == 2 and ck['meta']['iter'] == 6 and set(ck) == {'meta', 'state_dict', 'optimizer'}
    assert all(not k.startswith('module.') for k in ck['state_dict'])
    for (n, v), w in zip(ck['state_dict'].items(), w0):
        assert (v.numpy() == w).all(), n


def test_resume_continues_the_saved_run_and_load_from_takes_the_weights(tmp_path):
    """Save after 3 steps, resume into a fresh model: counters, schedules, optimizer moments and weights are those of the
    saved run, so the 4th step is bit-identical; ``load_checkpoint`` (load_from) takes the weights only."""
    data = lambda i: dict(points=synthetic.make_batch(2, start=10 * i, n_points=200, n_obj_range=(2, 3), n_ibp_range=(5, 10))['points'],
                          img_metas=[{}, {}])
    cfg = Config(CFG)
    a = Runner(_tiny(), cfg, max_iters=10)
    for i in range(3):
        a.step(data(i))
    a.epoch = 1
    path = a.save_checkpoint(str(tmp_path / 'w'))
    assert path.endswith('epoch_1.pth')
    out_a = a.step(data(3))
    b = Runner(_tiny(), cfg, max_iters=10)
    with torch.no_grad():
        for p in b.raw_model.parameters():
            p.add_(1.0)                                             # start from somewhere else
    meta = b.resume(str(tmp_path / 'w' / 'latest.pth'))
    assert meta['iter'] == 3 and b.iter == 3 and b.epoch == 1
    out_b = b.step(data(3))
    assert float(out_a['loss']) == float(out_b['loss'])
    for p, q in zip(a.raw_model.parameters(), b.raw_model.parameters()):
        assert torch.equal(p, q)
    assert a.optimizer.param_groups[0]['lr'] == b.optimizer.param_groups[0]['lr']
    c = Runner(_tiny(), cfg, max_iters=10)
    c.load_checkpoint(path)
    assert c.iter == 0 and c.epoch == 0
    sd = torch.load(path, weights_only=False)['state_dict']
    for n, p in c.raw_model.state_dict().items():
        assert torch.equal(p, sd[n])
    # a checkpoint written by a wrapped model ('module.' names) loads as well
    torch.save(dict(state_dict={'module.' + k: v for k, v in sd.items()}, meta=dict(epoch=0, iter=0)), str(tmp_path / 'wrapped.pth'))
    c.load_checkpoint(str(tmp_path / 'wrapped.pth'), strict=True)


def test_resume_with_the_step_policy_keeps_the_base_rates(tmp_path):
    """lr_config policy='step' (configs/gga/gga_pdg.py) with paramwise multipliers and warm-up: a run saved past its first decay
    milestone and resumed continues with the rates of the uninterrupted run - the base rates ride in the optimizer's param
    groups as mmcv's 'initial_lr', not re-derived from the checkpoint's already-decayed 'lr' (ADVICE r04)."""
    data = lambda i: dict(points=synthetic.make_batch(2, start=10 * i, n_points=200, n_obj_range=(2, 3), n_ibp_range=(5, 10))['points'],
                          img_metas=[{}, {}])
    cfg = Config(dict(CFG, optimizer=dict(type='SGD', lr=1e-3, momentum=0.9, weight_decay=1e-4,
                                          paramwise_cfg=dict(bias_lr_mult=2.0, bias_decay_mult=0.0)),
                      lr_config=dict(policy='step', step=[2, 4], gamma=0.1, warmup='linear', warmup_iters=2, warmup_ratio=1.0 / 3),
                      momentum_config=None))
    a = Runner(_tiny(), cfg, max_iters=12, iters_per_epoch=2)           # milestones at iterations 4 and 8
    want = []
    for i in range(10):
        if i == 5:
            a.epoch = 2
            path = a.save_checkpoint(str(tmp_path / 'w'))
        a.step(data(i))
        want.append([g['lr'] for g in a.optimizer.param_groups])
    assert want[0][0] == pytest.approx(1e-3 / 3) and want[3][0] == pytest.approx(1e-3) and want[4][0] == pytest.approx(1e-4)
    assert want[9][0] == pytest.approx(1e-5) and sorted(want[3]) == pytest.approx([1e-3, 2e-3])
    b = Runner(_tiny(), cfg, max_iters=12, iters_per_epoch=2)
    b.resume(path)
    assert b.iter == 5
    for i in range(5, 10):
        b.step(data(i))
        assert [g['lr'] for g in b.optimizer.param_groups] == want[i], i
    for p, q in zip(a.raw_model.parameters(), b.raw_model.parameters()):
        assert torch.equal(p, q)
    # a checkpoint of a run on three planes after a range-guard fall-back is resumed on three planes, and guarded on its first step
    from gga_amd import dense_conv
    ck = torch.load(path, weights_only=False)
    assert ck['meta']['gga_amd_planes'] == a.planes and ck['meta']['gga_amd_fell_back'] is False
    ck['meta']['gga_amd_fell_back'] = True
    torch.save(ck, str(tmp_path / 'fell_back.pth'))
    was = dense_conv.FELL_BACK
    try:
        c = Runner(_tiny(), cfg, max_iters=12, iters_per_epoch=2)
        c._guard_next = False
        c.resume(str(tmp_path / 'fell_back.pth'))
        assert c._guard_next and (c.planes == 3 or dense_conv.PLANES_PINNED)
    finally:
        dense_conv.FELL_BACK = was


def test_sampler_seed_is_common_to_the_ranks_and_one_process_drives_one_gpu():
    """tools/train.py --diff-seed offsets the generators of a rank, never the sampler's seed (mmdet: sync_random_seed): the
    shards stay disjoint. The non-distributed multi-GPU mode of the reference does not exist here: refused, not truncated."""
    ds = _Frames(16)
    loaders = [LD.build_dataloader(ds, 2, 0, dist=True, seed=7, worker_seed=7 + r, rank=r, world_size=2) for r in range(2)]
    a, b = (list(l.sampler) for l in loaders)
    assert not (set(a) & set(b)) and set(a) | set(b) == set(range(16))
    with pytest.raises(NotImplementedError):
        LD.build_dataloader(ds, 2, 0, num_gpus=2, dist=False)


# ----------------------------------------------------------------------------------------------------------------- device
@pytest.mark.gpu
def test_detector_trains_from_the_kitti_tree_saves_and_resumes(tmp_path):
    """On-disk KITTI tree -> KittiDataset_GGA_train -> train pipeline -> loader -> collate -> Runner steps of the PointPillars
    detector on the device -> checkpoint -> a fresh detector resumed from it repeats the next step bit for bit."""
    from gga_amd import build_model
    from gga_amd.cnn import to_channels_last
    DEV = torch.device('cuda:0')
    infos = kitti_tree(str(tmp_path))
    ds = LD.build_dataset(dataset_cfg(str(tmp_path), infos, times=3))
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
    cfg.model.pts_middle_encoder['channels_last'] = True
    cfg.data = dict(samples_per_gpu=3, workers_per_gpu=0)
    cfg.runner = dict(type='EpochBasedRunner', max_epochs=1)
    cfg.work_dir, cfg.seed = str(tmp_path / 'work'), 0
    cfg.checkpoint_config = dict(interval=1)

    def fresh():
        torch.manual_seed(0)
        model = build_model(cfg.model)
        with torch.no_grad():          # random init only: the raw Kaiming output convs decode boxes of e^10 m (inf / NaN losses)
            for th in model.pts_bbox_head.task_heads:
                for name in ('reg', 'height', 'dim', 'rot'):
                    getattr(th, name)[-1].weight.mul_(0.05)
        return to_channels_last(model.to(DEV)).train()

    np.random.seed(0), torch.manual_seed(0)
    model = fresh()
    runner = train_detector(model, ds, cfg, distributed=False, device=DEV)
    assert runner.iter == 3 and runner.epoch == 1 and os.path.exists(os.path.join(cfg.work_dir, 'epoch_1.pth'))
    # one more step on a fixed batch, from the live runner and from a resumed one
    np.random.seed(5), torch.manual_seed(5)
    batch = LD.to_step_inputs(next(iter(LD.build_dataloader(ds, 3, 0, dist=False, seed=0))), DEV)
    torch.manual_seed(9)                    # the SRL draws of the step come from the CPU generator
    out_a = runner.step(batch)
    assert set(out_a['log_vars']) >= {'task0.loss_heatmap', 'task2.distancemin', 'loss'} and np.isfinite(float(out_a['loss']))
    model_b = fresh()
    with torch.no_grad():
        for p in model_b.parameters():
            p.mul_(0.5)
    runner_b = Runner(model_b, cfg, max_iters=3, device=DEV)
    runner_b.resume(find_latest_checkpoint(cfg.work_dir))
    assert runner_b.iter == 3
    torch.manual_seed(9)
    out_b = runner_b.step(batch)
    assert float(out_a['loss']) == float(out_b['loss'])
    for (n, p), q in zip(model.state_dict().items(), model_b.state_dict().values()):
        assert torch.equal(p, q), n


# ------------------------------------------------------------------------------------------------- test / pseudo-label run
def matching_cfg(root, infos, model_cfg_path):
    """configs/gga/gga_kitti_matching_config.py's test section pointed at the synthetic tree."""
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_matching_config.py'))
    model = Config.fromfile(model_cfg_path)
    cfg.model, cfg.test_cfg = model.model, None
    rng = list(model.model.pts_voxel_layer.point_cloud_range)
    test = dict(cfg.data['test'])
    pipe = copy.deepcopy(list(test['pipeline']))
    for t in pipe[1]['transforms']:
        if t['type'] == 'PointsRangeFilter':
            t['point_cloud_range'] = rng
    test.update(data_root=root, ann_file=infos, pts_prefix='velodyne', pipeline=pipe, pcd_limit_range=rng)
    cfg.data = dict(samples_per_gpu=2, workers_per_gpu=0, test=test, test_dataloader=dict(samples_per_gpu=2, workers_per_gpu=0))
    return cfg


def test_test_pipeline_and_single_gpu_test_layout(tmp_path):
    """The reference's test_pipeline (MultiScaleFlipAug3D around the identity augmentations) on the tree: one augmentation,
    points untouched but for the range filter, and the list-over-augmentations-of-list-over-frames layout forward_test takes."""
    from gga_amd.apis import single_gpu_test
    infos = kitti_tree(str(tmp_path))
    cfg = matching_cfg(str(tmp_path), infos, os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
    ds = LD.build_dataset(dict(cfg.data['test'], test_mode=True))
    assert type(ds).__name__ == 'KittiDataset_GGA_match' and len(ds) == 3
    s = ds[0]
    assert isinstance(s['points'], list) and len(s['points']) == 1 and isinstance(s['points'][0], DC)
    raw = torch.from_numpy(np.fromfile(os.path.join(str(tmp_path), 'training', 'velodyne', f'{SEEDS[0]:06d}.bin'), dtype=np.float32).reshape(-1, 4))
    lo, hi = torch.tensor(PP_RANGE[:3]), torch.tensor(PP_RANGE[3:])
    keep = ((raw[:, :3] > lo) & (raw[:, :3] < hi)).all(1)
    assert torch.equal(s['points'][0].data, raw[keep])
    meta = s['img_metas'][0].data
    assert meta['sample_idx'] == SEEDS[0] and meta['pcd_scale_factor'] == 1.0 and not meta['pcd_horizontal_flip'] and not meta['flip']
    seen = []

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, return_loss=True, rescale=False, points=None, img_metas=None):
            assert not return_loss and rescale and len(points) == len(img_metas) == 1
            seen.append([m['sample_idx'] for m in img_metas[0]])
            return [dict(pts_bbox=dict(n=len(p))) for p in points[0]]

    loader = LD.build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=0, dist=False, shuffle=False)
    out = single_gpu_test(Stub(), loader, torch.device('cpu'))
    assert seen == [[SEEDS[0], SEEDS[1]], [SEEDS[2]]] and len(out) == 3 and out[0]['pts_bbox']['n'] == int(keep.sum())


@pytest.mark.gpu
def test_pseudo_label_generation_run(tmp_path):
    """tools/generate_pseudo_labels_gga.py's flow on the tree: checkpoint -> detector -> detections of every frame ->
    KittiDataset_GGA_match.evaluate -> the pseudo-label file (one re-labelled info per frame, GGA fields following the match)."""
    from gga_amd import build_model
    from gga_amd.apis import generate_pseudo_labels
    infos = kitti_tree(str(tmp_path))
    model_cfg = os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py')
    cfg = matching_cfg(str(tmp_path), infos, model_cfg)
    torch.manual_seed(0)
    model = build_model(Config.fromfile(model_cfg).model)
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
            th.heatmap[-1].bias.fill_(0.5)                 # a random-init detector that reports boxes
    ck = str(tmp_path / 'epoch_1.pth')
    torch.save(dict(meta=dict(epoch=1, iter=3, CLASSES=('Pedestrian', 'Cyclist', 'Car')),
                    state_dict={'module.' + k: v for k, v in model.state_dict().items()}), ck)
    out_file = str(tmp_path / 'pseudo.pkl')
    outputs, res = generate_pseudo_labels(cfg, ck, out=str(tmp_path / 'raw.pkl'), eval_metrics=('mAP',),
                                          eval_options=dict(pseudo_label_file=out_file))
    assert len(outputs) == 3 and all(set(o['pts_bbox']) >= {'boxes_3d', 'scores_3d', 'labels_3d'} for o in outputs)
    assert sum(len(o['pts_bbox']['scores_3d']) for o in outputs) > 0
    assert res['pseudo_labels/frames'] == 3.0 and os.path.exists(str(tmp_path / 'raw.pkl'))
    labelled = pickle.load(open(out_file, 'rb'))
    assert len(labelled) == 3
    for info, src in zip(labelled, infos):
        a = info['annos']
        assert info['image']['image_idx'] == src['image']['image_idx']
        n = len(a['name'])
        assert all(len(a[k]) == n for k in ('bbox', 'dimensions', 'location', 'rotation_y', 'score', 'GGA_boxes_img', 'GGA_init_pseudo_label'))
        assert (a['dimensions'][:, 0] >= a['dimensions'][:, 2]).all()      # longer horizontal side first
    assert sum(len(i['annos']['name']) for i in labelled) == res['pseudo_labels/detections']


@pytest.mark.gpu
@pytest.mark.parametrize('collect', ['cpu', 'gpu'])
def test_pseudo_label_generation_over_two_ranks(collect, tmp_path):
    """tools/dist_pseudo.sh's flow (reference: tools/generate_pseudo_labels_gga.py:236-262 with ``multi_gpu_test``): two ranks
    (gloo, sharing the box's one GPU) test strided shards of the three frames, rank 0 collects - through files or through the
    process group -, writes the raw outputs and the pseudo-label file: frame for frame the detections of the single-process run."""
    import subprocess
    import sys
    from gga_amd import build_model
    from gga_amd.apis import generate_pseudo_labels
    infos = kitti_tree(str(tmp_path))
    model_cfg = os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py')
    torch.manual_seed(0)
    model = build_model(Config.fromfile(model_cfg).model)
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
            th.heatmap[-1].bias.fill_(0.5)
    ck = str(tmp_path / 'epoch_1.pth')
    torch.save(dict(meta=dict(epoch=1, iter=3, CLASSES=('Pedestrian', 'Cyclist', 'Car')), state_dict=model.state_dict()), ck)
    cfg = matching_cfg(str(tmp_path), infos, model_cfg)
    single, _ = generate_pseudo_labels(cfg, ck, eval_metrics=('mAP',), eval_options=dict(pseudo_label_file=str(tmp_path / 'single.pkl')))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GGA_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    raw, out_file = str(tmp_path / 'raw.pkl'), str(tmp_path / 'pseudo.pkl')
    run = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(REPO, 'tests', '_pseudo_dist_worker.py'), str(tmp_path), ck, raw, out_file,
                          collect], env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and 'PSEUDO frames 3' in run.stdout, (run.stdout[-1500:], run.stderr[-2500:])
    both = pickle.load(open(raw, 'rb'))
    assert len(both) == len(single) == 3
    for a, b in zip(both, single):
        assert torch.equal(a['pts_bbox']['boxes_3d'].tensor, b['pts_bbox']['boxes_3d'].tensor)
        assert torch.equal(a['pts_bbox']['scores_3d'], b['pts_bbox']['scores_3d']) and torch.equal(a['pts_bbox']['labels_3d'], b['pts_bbox']['labels_3d'])
    relabelled, alone = pickle.load(open(out_file, 'rb')), pickle.load(open(str(tmp_path / 'single.pkl'), 'rb'))
    assert [i['image']['image_idx'] for i in relabelled] == [i['image']['image_idx'] for i in alone]
    for x, y in zip(relabelled, alone):
        assert all(np.array_equal(x['annos'][k], y['annos'][k]) for k in ('name', 'bbox', 'location', 'GGA_init_pseudo_label'))


@pytest.mark.gpu
def test_whole_recipe_on_the_synthetic_tree(tmp_path):
    """The GGA recipe end to end (reference README.md:159-192) on the three-frame tree: GT database from the info file ->
    the reference's train_pipeline WITH database sampling -> tools/train.py's flow (train_detector, one epoch, checkpoint) ->
    tools/generate_pseudo_labels_gga.py's flow with that checkpoint -> a pseudo-label info file the train dataset loads again
    (the retraining step of the recipe reads it as its ann_file)."""
    from gga_amd import build_model
    from gga_amd.apis import generate_pseudo_labels
    from gga_amd.cnn import to_channels_last
    from gga_amd.datasets import KittiDataset_GGA_train, LoadAnnotations3D
    from gga_amd.gt_database import create_groundtruth_database
    from gga_amd.pipelines import LoadPointsFromFile
    DEV = torch.device('cuda:0')
    root = str(tmp_path)
    infos = kitti_tree(root)
    # 1. GT database (tools/create_data_gga.py kitti -> create_groundtruth_database)
    plain = KittiDataset_GGA_train(root, infos, 'training', classes=CLASSES, modality=dict(use_lidar=True, use_camera=False),
                                   pipeline=[LoadPointsFromFile(coord_type='LIDAR', load_dim=4, use_dim=4),
                                             LoadAnnotations3D(with_bbox_3d=True, with_label_3d=True)])
    db = create_groundtruth_database(plain, root, 'kitti', logger=lambda s: None)
    assert os.path.exists(os.path.join(root, 'kitti_dbinfos_train_GGA.pkl')) and sum(len(v) for v in db.values()) > 0
    # 2. training from the dataset with the reference's train_pipeline (database sampling included)
    model_cfg = os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py')
    cfg = Config.fromfile(model_cfg)
    pipe = train_pipeline()
    groups = {k: 2 for k in db if k in CLASSES}
    pipe.insert(2, dict(type='ObjectSample_GGA', min_distance=5.0,
                        db_sampler=dict(data_root=root, info_path=os.path.join(root, 'kitti_dbinfos_train_GGA.pkl'), rate=1.0,
                                        prepare=dict(filter_by_difficulty=[-1], filter_by_min_points={k: 1 for k in groups}),
                                        classes=CLASSES, sample_groups=groups)))
    ds_cfg = dataset_cfg(root, infos, times=2)
    ds_cfg['dataset']['pipeline'] = pipe
    cfg.model.pts_middle_encoder['channels_last'] = True
    cfg.data = dict(samples_per_gpu=3, workers_per_gpu=0, train=ds_cfg)
    cfg.runner, cfg.checkpoint_config = dict(type='EpochBasedRunner', max_epochs=1), dict(interval=1)
    cfg.work_dir, cfg.seed = os.path.join(root, 'work'), 0
    np.random.seed(0), torch.manual_seed(0)
    model = build_model(cfg.model)
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
            th.heatmap[-1].bias.fill_(0.5)
    model = to_channels_last(model.to(DEV)).train()
    ds = LD.build_dataset(cfg.data['train'])
    sample = ds[0]
    assert len(sample['gt_labels_3d'].data) >= 1                      # objects (own + pasted ones) survive the filters
    runner = train_detector(model, ds, cfg, distributed=False, device=DEV)
    assert runner.iter == 2 and runner.epoch == 1
    ck = os.path.join(cfg.work_dir, 'epoch_1.pth')
    assert os.path.exists(ck)
    # 3. pseudo-label generation with the trained checkpoint
    mcfg = matching_cfg(root, infos, model_cfg)
    out_file = os.path.join(root, 'kitti_infos_trainval_GGA_pseudo.pkl')
    outputs, res = generate_pseudo_labels(mcfg, ck, eval_metrics=('mAP',), eval_options=dict(pseudo_label_file=out_file))
    assert len(outputs) == 3 and res['pseudo_labels/frames'] == 3.0 and os.path.exists(out_file)
    # 4. the retraining step reads that file as its ann_file: the infos keep the layout the datasets take
    relabelled = pickle.load(open(out_file, 'rb'))
    assert [i['image']['image_idx'] for i in relabelled] == [i['image']['image_idx'] for i in infos]
    for info in relabelled:
        assert {'name', 'bbox', 'dimensions', 'location', 'rotation_y', 'GGA_boxes_img', 'GGA_init_pseudo_label'} <= set(info['annos'])
