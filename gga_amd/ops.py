"""Post-processing ops of the inference / pseudo-label path on the HIP kernels of
gga_amd/csrc/postproc.hip, with the call signatures the reference uses:

* ``nms_rotated(boxes_xywhr, scores, iou_threshold) -> (dets, keep)`` — ``mmcv.ops.nms_rotated``
  (mmdet3d/core/post_processing/box3d_nms.py:5,264)
* ``nms_bev(boxes_xyxyr, scores, thresh, pre_max_size, post_max_size)`` — box3d_nms.py:231-268
* ``box_iou_rotated(b1, b2, mode='iou', aligned=False)`` — mmcv op used by base_box3d.py:469
* ``points_in_boxes_part / points_in_boxes_all(points [B,M,3], boxes [B,T,7])`` — base_box3d.py:534,566
* ``xywhr2xyxyr`` — mmdet3d/core/bbox/structures/utils.py:121-139
"""
import torch

from . import _lib
from . import functional as F
from ._lib import check


def xywhr2xyxyr(boxes_xywhr):
    """(centre x, centre y, w, h, r) -> (x1, y1, x2, y2, r): the corners of the axis-aligned box before rotation, one
    ``cat`` instead of five strided writes (same values: centre -/+ extent / 2)."""
    centre, half, rot = boxes_xywhr[..., 0:2], boxes_xywhr[..., 2:4] / 2, boxes_xywhr[..., 4:5]
    return torch.cat([centre - half, centre + half, rot], dim=-1)


@torch.no_grad()
def box_iou_rotated(bboxes1, bboxes2, mode='iou', aligned=False):
    assert mode in ('iou', 'iof')
    F._need_cuda(bboxes1, bboxes2)
    b1, b2 = bboxes1.float().contiguous(), bboxes2.float().contiguous()
    n, m = b1.shape[0], b2.shape[0]
    out = torch.zeros((n,) if aligned else (n, m), dtype=torch.float32, device=b1.device)
    check(_lib.lib().gga_box_iou_rotated(F._p(b1), n, F._p(b2), m, int(mode == 'iof'), int(aligned), F._p(out),
                                         F._stream()), 'gga_box_iou_rotated')
    return out


@torch.no_grad()
def nms_rotated(dets, scores, iou_threshold, labels=None, max_keep=0):
    """dets [N,5] (x, y, w, h, angle) -> (cat(dets[keep], scores[keep]), keep). ``keep`` is ordered
    by descending score. Multi-label NMS (``labels``) offsets the boxes per label like mmcv."""
    F._need_cuda(dets, scores)
    if dets.shape[0] == 0:
        return dets, dets.new_zeros(0, dtype=torch.long)
    boxes = dets.float()
    if labels is not None:       # boxes of different labels never overlap
        span = boxes[:, :2].abs().max() + boxes[:, 2:4].max() + 1
        boxes = boxes.clone()
        boxes[:, :2] += labels.to(boxes)[:, None] * span * 2
    scores_sorted, order = scores.sort(0, descending=True)
    boxes_sorted = boxes.index_select(0, order).contiguous()
    n = boxes_sorted.shape[0]
    L = _lib.lib()
    keep_pos = torch.empty(n, dtype=torch.int64, device=dets.device)
    num = torch.zeros(1, dtype=torch.int32, device=dets.device)
    ws = F._workspace('nms', L.gga_nms_rotated_workspace_bytes(n), dets.device)
    check(L.gga_nms_rotated_sorted(F._p(boxes_sorted), n, float(iou_threshold), int(max_keep), F._p(keep_pos), F._p(num),
                                   F._p(ws), ws.numel(), F._stream()), 'gga_nms_rotated_sorted')
    keep = order[keep_pos[:int(num.item())]]
    return torch.cat([dets[keep], scores[keep].reshape(-1, 1)], dim=1), keep


def circle_nms(dets, thresh, post_max_size=83):
    """Circular NMS (mmdet3d/core/post_processing/box3d_nms.py:181-225) on the device: dets [N,3] =
    (x, y, score); a centre survives if no higher-scored kept centre lies within squared distance
    ``thresh``. Returns the kept indices by descending score, at most ``post_max_size``. (The
    reference sorts with numpy's quicksort, which leaves the order of equal scores unspecified;
    here ties keep their input order.)"""
    F._need_cuda(dets)
    n = dets.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=dets.device)
    order = torch.sort(dets[:, 2].float(), descending=True, stable=True)[1]
    xy = dets[:, :2].float().index_select(0, order).contiguous()
    L = _lib.lib()
    keep_pos = torch.empty(n, dtype=torch.int64, device=dets.device)
    num = torch.zeros(1, dtype=torch.int32, device=dets.device)
    ws = F._workspace('nms', L.gga_nms_rotated_workspace_bytes(n), dets.device)
    check(L.gga_circle_nms_sorted(F._p(xy), n, float(thresh), int(post_max_size or 0), F._p(keep_pos), F._p(num), F._p(ws),
                                  ws.numel(), F._stream()), 'gga_circle_nms_sorted')
    return order[keep_pos[:int(num.item())]]


def nms_bev(boxes, scores, thresh, pre_max_size=None, post_max_size=None):
    assert boxes.size(1) == 5, 'Input boxes shape should be [N, 5]'
    order = scores.sort(0, descending=True)[1]
    if pre_max_size is not None:
        order = order[:pre_max_size]
    boxes = boxes[order].contiguous()
    scores = scores[order]
    # xyxyr -> xywhr (box3d_nms.py:258-262)
    boxes = torch.stack(((boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2,
                         boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1], boxes[:, 4]), dim=-1)
    keep = nms_rotated(boxes, scores, thresh, max_keep=post_max_size or 0)[1]
    keep = order[keep]
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep


@torch.no_grad()
def _points_in_boxes(points, boxes, all_boxes):
    F._need_cuda(points, boxes)
    assert points.dim() == 3 and boxes.dim() == 3 and points.shape[0] == boxes.shape[0], \
        f'points {tuple(points.shape)} / boxes {tuple(boxes.shape)} must be [B,M,3] / [B,T,7]'
    assert points.shape[2] == 3 and boxes.shape[2] == 7
    B, M, _ = points.shape
    T = boxes.shape[1]
    pts, bx = points.float().contiguous(), boxes.float().contiguous()
    out = torch.zeros((B, M, T) if all_boxes else (B, M), dtype=torch.int32, device=points.device)
    if not all_boxes:
        out.fill_(-1)
    check(_lib.lib().gga_points_in_boxes(F._p(pts), F._p(bx), B, M, T, int(all_boxes), F._p(out), F._stream()),
          'gga_points_in_boxes')
    return out


def points_in_boxes_part(points, boxes):
    return _points_in_boxes(points, boxes, False)


def points_in_boxes_all(points, boxes):
    return _points_in_boxes(points, boxes, True)


def nms_normal_bev(boxes, scores, thresh):
    """core/post_processing/box3d_nms.py:274-288: NMS of the boxes with their yaw set to 0 (axis-aligned
    xyxy boxes) - the rotated-NMS kernel with angle 0."""
    assert boxes.shape[1] == 5, 'Input boxes shape should be [N, 5]'
    b = torch.stack(((boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2, boxes[:, 2] - boxes[:, 0],
                     boxes[:, 3] - boxes[:, 1], torch.zeros_like(boxes[:, 0])), dim=-1)
    return nms_rotated(b, scores, thresh)[1]


def box3d_multiclass_nms(mlvl_bboxes, mlvl_bboxes_for_nms, mlvl_scores, score_thr, max_num, cfg, mlvl_dir_scores=None,
                         mlvl_attr_scores=None, mlvl_bboxes2d=None):
    """core/post_processing/box3d_nms.py:8-127: per-class BEV NMS of the score-thresholded boxes (rotated or
    axis-aligned on the device), then the ``max_num`` best overall. ``mlvl_scores`` [N, C + 1] (last column =
    background padding). -> (bboxes, scores, labels, dir_scores[, attr_scores][, bboxes2d])."""
    num_classes = mlvl_scores.shape[1] - 1
    use_rotate = cfg['use_rotate_nms'] if isinstance(cfg, dict) else cfg.use_rotate_nms
    nms_thr = cfg['nms_thr'] if isinstance(cfg, dict) else cfg.nms_thr
    picks = []
    for i in range(num_classes):
        cls_inds = mlvl_scores[:, i] > score_thr
        if not cls_inds.any():
            continue
        idx = cls_inds.nonzero().squeeze(1)
        sc = mlvl_scores[idx, i]
        selected = (nms_bev if use_rotate else nms_normal_bev)(mlvl_bboxes_for_nms[idx], sc, nms_thr)
        picks.append((idx[selected], sc[selected], mlvl_bboxes.new_full((len(selected), ), i, dtype=torch.long)))
    extras = [t for t in (mlvl_dir_scores, mlvl_attr_scores, mlvl_bboxes2d) if t is not None]
    if picks:
        idx = torch.cat([p[0] for p in picks])
        scores, labels = torch.cat([p[1] for p in picks]), torch.cat([p[2] for p in picks])
        if idx.shape[0] > max_num:
            inds = scores.sort(descending=True)[1][:max_num]
            idx, scores, labels = idx[inds], scores[inds], labels[inds]
        return (mlvl_bboxes[idx], scores, labels) + tuple(t[idx] for t in extras)
    empty = (mlvl_scores.new_zeros((0, mlvl_bboxes.size(-1))), mlvl_scores.new_zeros((0, )),
             mlvl_scores.new_zeros((0, ), dtype=torch.long))
    return empty + tuple(t.new_zeros((0, ) + tuple(t.shape[1:])) for t in extras)
