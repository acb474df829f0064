"""§8(f) rank 1 ops on the GPU: the reference's known-answer tests (tests/test_utils/test_nms.py:82-98,
test_box3d.py:1122-1158,1683-1745) through the product API, plus randomized parity with the oracle,
and the inference path (simple_test) of the detector."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO
from gga_amd import Config, build_model, ops, synthetic
from gga_amd.box3d import LiDARInstance3DBoxes
from oracle import oracle as O
from test_oracle import (KA_ALL, KA_BOXES, KA_DEPTH_BOXES, KA_DEPTH_PART, KA_DEPTH_PTS, KA_IOF3D, KA_IOU3D,
                         KA_NMS_BOXES, KA_NMS_SCORES, KA_OV1, KA_OV2, KA_PART, KA_PTS)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.asarray(a)).to(DEV)


def test_reference_known_answers():
    boxes = LiDARInstance3DBoxes(T(KA_BOXES))
    assert boxes.points_in_boxes_all(T(KA_PTS)).cpu().tolist() == KA_ALL
    assert boxes.points_in_boxes_part(T(KA_PTS)).cpu().tolist() == KA_PART
    assert LiDARInstance3DBoxes(T(KA_DEPTH_BOXES)).points_in_boxes_part(T(KA_DEPTH_PTS)[None]).cpu().tolist() == KA_DEPTH_PART
    assert ops.nms_bev(T(KA_NMS_BOXES), T(KA_NMS_SCORES), thresh=0.3).cpu().tolist() == [1, 0, 3]
    b1, b2 = LiDARInstance3DBoxes(T(KA_OV1)), LiDARInstance3DBoxes(T(KA_OV2))
    torch.testing.assert_close(LiDARInstance3DBoxes.overlaps(b1, b2).cpu(), torch.from_numpy(KA_IOU3D), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(LiDARInstance3DBoxes.overlaps(b1, b2, mode='iof').cpu(), torch.from_numpy(KA_IOF3D),
                               rtol=1e-3, atol=1e-4)
    assert LiDARInstance3DBoxes.overlaps(LiDARInstance3DBoxes([]), b2).shape == (0, 4)


def _rand_boxes(n, seed, spread=20.0):
    g = np.random.default_rng(seed)
    return np.stack([g.uniform(-spread, spread, n), g.uniform(-spread, spread, n), g.uniform(0.5, 6, n),
                     g.uniform(0.5, 6, n), g.uniform(-4, 4, n)], 1).astype(np.float32)


def test_box_iou_rotated_vs_oracle():
    b1, b2 = _rand_boxes(150, 1, 6), _rand_boxes(170, 2, 6)
    b2[:10] = b1[:10]                                        # identical boxes -> IoU 1
    for mode in ('iou', 'iof'):
        got = ops.box_iou_rotated(T(b1), T(b2), mode=mode).cpu().numpy()
        np.testing.assert_allclose(got, O.box_iou_rotated(b1, b2, mode), rtol=1e-4, atol=2e-5)
    al = ops.box_iou_rotated(T(b1), T(b2[:150]), aligned=True).cpu().numpy()
    np.testing.assert_allclose(al[:10], 1.0, atol=1e-5)


@pytest.mark.parametrize('n,thr', [(1, 0.3), (63, 0.1), (64, 0.5), (500, 0.2), (4096, 0.3)])
def test_nms_rotated_vs_oracle(n, thr):
    boxes = _rand_boxes(n, n, spread=4.0 * max(1.0, (n / 50) ** 0.5))
    scores = np.random.default_rng(n + 1).permutation(n).astype(np.float32) / n      # distinct scores
    dets, keep = ops.nms_rotated(T(boxes), T(scores), thr)
    ref = O.nms_rotated(boxes, scores, thr)
    got = keep.cpu().numpy()
    if not np.array_equal(got, ref):
        # a decision may flip only where an IoU sits within fp32 noise of the threshold
        iou = O.box_iou_rotated(boxes, boxes)
        assert (np.abs(iou - thr) < 1e-5).any(), (len(got), len(ref))
    assert dets.shape == (len(got), 6) and torch.equal(dets[:, 5].cpu(), torch.from_numpy(scores[got]))
    assert ops.nms_rotated(T(boxes), T(scores), thr, max_keep=3)[1].numel() == min(3, len(ref))


def test_points_in_boxes_batch_vs_oracle():
    g = np.random.default_rng(3)
    B, M, Tn = 3, 5000, 40
    pts = g.uniform(-12, 12, (B, M, 3)).astype(np.float32)
    boxes = np.concatenate([g.uniform(-10, 10, (B, Tn, 2)), g.uniform(-3, 1, (B, Tn, 1)), g.uniform(0.5, 6, (B, Tn, 3)),
                            g.uniform(-4, 4, (B, Tn, 1))], 2).astype(np.float32)
    part = ops.points_in_boxes_part(T(pts), T(boxes)).cpu().numpy()
    allb = ops.points_in_boxes_all(T(pts), T(boxes)).cpu().numpy()
    for b in range(B):
        assert np.array_equal(part[b], O.points_in_boxes(pts[b], boxes[b]))
        assert np.array_equal(allb[b], O.points_in_boxes(pts[b], boxes[b], all_boxes=True))
    assert (part >= 0).sum() > 100


def test_simple_test_inference_path():
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
    torch.manual_seed(0)
    model = build_model(cfg.model).to(DEV).eval()
    b = synthetic.make_batch(2, n_points=8000, pc_range=synthetic.RANGE_PP)
    res = model(return_loss=False, points=[[p.to(DEV) for p in b['points']]], img_metas=[b['img_metas']])
    assert len(res) == 2
    tc = cfg.model.test_cfg.pts
    for r in res:
        d = r['pts_bbox']
        n = len(d['boxes_3d'])
        assert d['boxes_3d'].tensor.shape == (n, 7) and d['scores_3d'].shape == (n,) and d['labels_3d'].shape == (n,)
        assert n <= 3 * tc.post_max_size and set(d['labels_3d'].tolist()) <= {0, 1, 2}
        if n:
            assert float(d['scores_3d'].min()) >= tc.score_threshold
            # survivors of one class do not overlap beyond the NMS threshold
            for c in range(3):
                bx = d['boxes_3d'].tensor[d['labels_3d'] == c].to(DEV)
                if len(bx) > 1:
                    iou = ops.box_iou_rotated(bx[:, [0, 1, 3, 4, 6]], bx[:, [0, 1, 3, 4, 6]]).cpu()
                    iou.fill_diagonal_(0)
                    assert float(iou.max()) <= tc.nms_thr + 1e-4


def test_circle_nms_known_answer_and_vs_oracle():
    """Reference known answer (tests/test_utils/test_nms.py:63-79) and the oracle's loop-for-loop
    restatement on 3000 random centres (clusters, exact-threshold pairs, duplicate centres)."""
    from gga_amd import ops
    from oracle import oracle as O
    boxes = torch.tensor([[-11.1100, 2.1300, 0.8823], [-11.2810, 2.2422, 0.8914], [-10.3966, -0.3198, 0.8643],
                          [-10.2906, -13.3159, 0.8401], [5.6518, 9.9791, 0.8271], [-11.2652, 13.3637, 0.8267],
                          [4.7768, -13.0409, 0.7810], [5.6621, 9.0422, 0.7753], [-10.5561, 18.9627, 0.7518],
                          [-10.5643, 13.2293, 0.7200]])
    keep = ops.circle_nms(boxes.to(DEV), 0.175, post_max_size=None)
    assert sorted(keep.tolist()) == [1, 2, 3, 4, 5, 6, 7, 8, 9] and int(keep[0]) == 1
    g = torch.Generator().manual_seed(0)
    ctr = torch.rand(40, 2, generator=g) * 60
    xy = ctr[torch.randint(0, 40, (3000,), generator=g)] + torch.randn(3000, 2, generator=g) * 1.5
    xy[100] = xy[7]                                   # duplicate centre
    xy[200] = xy[9] + torch.tensor([0.5, 0.0])        # distance^2 = 0.25
    dets = torch.cat([xy, torch.rand(3000, 1, generator=g)], 1)
    for thr, pm in ((0.25, 83), (4.0, 500), (0.01, None)):
        want = O.circle_nms(dets.numpy(), thr, pm)
        got = ops.circle_nms(dets.to(DEV), thr, pm).tolist()
        assert got == want, (thr, pm)
    assert ops.circle_nms(torch.zeros(0, 3, device=DEV), 1.0).numel() == 0


@pytest.mark.parametrize('cfg_name,B,dim_shift,bias', [('gga_kitti_pointpillars_config.py', 5, 1.2, -2.19), ('gga_kitti_config.py', 3, 0.3, -1.0),
                                                       ('gga_kitti_config.py', 2, 2.5, -6.0)])
def test_batched_detections_equal_the_per_frame_loop(cfg_name, B, dim_shift, bias, monkeypatch):
    """``CenterHead_GGA.get_bboxes`` (centerpoint_head_gga.py:725-934): the one-launch form for all frames and tasks
    (csrc/postproc.hip::centerpoint_detect_kernel) against the reference's per-(frame, task) loop - coder masks, score
    threshold, nms_bev on the xyxyr round trip, range filter, merge - which the known-answer tests above pin: the same
    detections in the same order, bit for bit (boxes, scores, labels), on random head outputs with enough overlap for the NMS
    to suppress (dim_shift), around the score threshold (bias), and with no survivor at all (bias -6)."""
    from gga_amd.dense_heads import CenterHead_GGA
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', cfg_name))
    torch.manual_seed(5)
    model = build_model(cfg.model).to(DEV).eval()
    head = model.pts_bbox_head
    W, H = head._feature_map_size() if head.train_cfg else (216, 248)
    g = torch.Generator(device='cpu').manual_seed(B)
    preds = []
    for t in range(len(head.task_heads)):
        r = lambda c, s=1.0: (torch.randn(B, c, H, W, generator=g) * s).to(DEV)
        preds.append([dict(heatmap=r(1, 0.6) + bias, reg=r(2, 0.3).sigmoid(), height=r(1), dim=r(3, 0.4) + dim_shift, rot=r(2))])
    metas = [dict(box_type_3d=LiDARInstance3DBoxes) for _ in range(B)]
    monkeypatch.setattr(CenterHead_GGA, 'BATCHED', False)
    loop = head.get_bboxes(preds, metas)
    monkeypatch.setattr(CenterHead_GGA, 'BATCHED', True)
    fast = head.get_bboxes(preds, metas)
    assert getattr(fast, 'packed', None) is not None and getattr(loop, 'packed', None) is None
    total = suppressed = 0
    for (b0, s0, l0), (b1, s1, l1) in zip(loop, fast):
        assert b0.tensor.shape == b1.tensor.shape, (b0.tensor.shape, b1.tensor.shape)
        assert torch.equal(b0.tensor, b1.tensor) and torch.equal(s0, s1) and torch.equal(l0.int(), l1.int())
        total += len(s0)
    if bias > -5:
        # the case is not vacuous: candidates over the threshold, and the NMS removed some of them
        cand = sum(int(((p[0]['heatmap'].sigmoid().reshape(B, -1).topk(head.bbox_coder.max_num)[0]) >= 0.1).sum()) for p in preds)
        assert 0 < total < cand, (total, cand)
    else:
        assert total == 0
    # and through the detector: simple_test's packed hand-over gives the per-frame result dicts of the loop
    model.pts_bbox_head.BATCHED = True


def test_simple_test_batched_results_equal_the_loop(monkeypatch):
    from gga_amd.dense_heads import CenterHead_GGA
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
    torch.manual_seed(0)
    model = build_model(cfg.model).to(DEV).eval()
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
            th.dim[-1].bias.fill_(1.0)
    b = synthetic.make_batch(3, n_points=6000, pc_range=synthetic.RANGE_PP)
    pts = [p.to(DEV) for p in b['points']]
    metas = [dict(m, box_type_3d=LiDARInstance3DBoxes) for m in b['img_metas']]
    monkeypatch.setattr(CenterHead_GGA, 'BATCHED', False)
    loop = model.simple_test(pts, metas)
    monkeypatch.setattr(CenterHead_GGA, 'BATCHED', True)
    fast = model.simple_test(pts, metas)
    assert sum(len(r['pts_bbox']['scores_3d']) for r in loop) > 0
    for r0, r1 in zip(loop, fast):
        a, c = r0['pts_bbox'], r1['pts_bbox']
        assert not c['boxes_3d'].tensor.is_cuda and type(c['boxes_3d']) is type(a['boxes_3d'])
        assert torch.equal(a['boxes_3d'].tensor, c['boxes_3d'].tensor) and torch.equal(a['scores_3d'], c['scores_3d'])
        assert torch.equal(a['labels_3d'].int(), c['labels_3d'].int())
