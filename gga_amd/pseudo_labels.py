"""Pseudo-label matching: the step after training in the GGA recipe (README "Retrain" stage).

Mirror of ``tools/utils_pseudo_labels_gga.py:11-84`` of the reference, called from
``KittiDataset_GGA_match.evaluate`` (``mmdet3d/datasets/kitti_dataset_GGA_match.py:421-424``):
every predicted 3D box (its projected 2D ``bbox``) is matched to the ground-truth 2D box of the
same frame with the largest image-plane IoU; the prediction's own fields replace the annotation
and the GGA side information (everything the detector does not predict) is taken from the
matched ground truth. The IoU + argmax of all frames run as ONE launch of
``gga_image_box_match`` (include/gga_hip.h); the dictionary bookkeeping is host code as in the
reference. Same names, arguments, in-place behaviour and return value as the reference.
"""
import copy
import os
import pickle

import numpy as np
import torch

from . import _lib
from . import functional as F
from ._lib import check

DEFAULT_OUT_FILE = './data/kitti_pesudo/kitti_infos_trainval_GGA_pseudo.pkl'   # utils_pseudo_labels_gga.py:71


def drop_arrays_by_name(gt_names, used_classes=['Pedestrian', 'Car', 'Cyclist']):
    """Indices of the entries whose name is one of ``used_classes`` (utils_pseudo_labels_gga.py:11-15)."""
    return np.array([i for i, n in enumerate(gt_names) if n in used_classes], dtype=np.int64)


def image_box_match(dt_bboxes, gt_bboxes, device='cuda:0', return_overlaps=False):
    """Per frame ``np.argmax(image_box_overlap(dt, gt), axis=-1)``.

    dt_bboxes / gt_bboxes: lists (one entry per frame) of [n,4] arrays (x1,y1,x2,y2). Overlaps
    are rounded to float32 when the detections are float32, as the reference's
    ``np.zeros((N, K), dtype=boxes.dtype)`` does (eval.py:89). Returns a list of int64 index
    arrays (and the list of overlap matrices in the detections' dtype).
    """
    n_frames = len(dt_bboxes)
    assert n_frames == len(gt_bboxes)
    if n_frames == 0:
        return ([], []) if return_overlaps else []
    dts = [np.asarray(b).reshape(-1, 4) for b in dt_bboxes]
    gts = [np.asarray(b).reshape(-1, 4) for b in gt_bboxes]
    dt_dtype = np.result_type(*[d.dtype for d in dts]) if dts else np.float64
    dt_off = np.zeros(n_frames + 1, np.int64)
    gt_off = np.zeros(n_frames + 1, np.int64)
    ov_off = np.zeros(n_frames + 1, np.int64)
    for f in range(n_frames):
        dt_off[f + 1] = dt_off[f] + len(dts[f])
        gt_off[f + 1] = gt_off[f] + len(gts[f])
        ov_off[f + 1] = ov_off[f] + len(dts[f]) * len(gts[f])
        if len(dts[f]) and not len(gts[f]):
            # np.argmax over an empty axis
            raise ValueError('attempt to get argmax of an empty sequence')
    n_dt = int(dt_off[-1])
    dev = torch.device(device)
    cat = lambda parts: np.concatenate(parts, 0).astype(np.float64) if parts else np.zeros((0, 4))
    d_dt = torch.from_numpy(cat(dts)).to(dev)
    d_gt = torch.from_numpy(cat(gts)).to(dev)
    d_dto, d_gto, d_ovo = (torch.from_numpy(a).to(dev) for a in (dt_off, gt_off, ov_off))
    match = torch.empty(n_dt, dtype=torch.int64, device=dev)
    ov = torch.empty(int(ov_off[-1]), dtype=torch.float64, device=dev) if return_overlaps else None
    F._need_cuda(d_dt)
    with torch.cuda.device(dev):
        check(_lib.lib().gga_image_box_match(F._p(d_dt), F._p(d_dto), F._p(d_gt), F._p(d_gto), n_frames, n_dt,
                                             int(dt_dtype == np.float32), F._p(match), None, F._p(ov) if ov is not None else None,
                                             F._p(d_ovo), F._stream()), 'gga_image_box_match')
    match = match.cpu().numpy()
    out = [match[dt_off[f]:dt_off[f + 1]] for f in range(n_frames)]
    if not return_overlaps:
        return out
    ov = ov.cpu().numpy()
    mats = [ov[ov_off[f]:ov_off[f + 1]].reshape(len(dts[f]), len(gts[f])).astype(dt_dtype) for f in range(n_frames)]
    return out, mats


def pseudo_label_matching_kitti(gt_infos, dt_annos, metric=0, num_parts=200, filename=DEFAULT_OUT_FILE, device='cuda:0'):
    """utils_pseudo_labels_gga.py:17-84. ``gt_infos``: KITTI info dicts (``info['annos']`` holds
    per-object arrays incl. the GGA fields); ``dt_annos``: KITTI-format detections. Writes the
    re-labelled infos to ``filename`` (pickle; None skips the dump) and returns the cleaned
    ``gt_annos`` (in-box points and DontCare / unused classes removed), modified in place as
    the reference does. ``num_parts`` only sized the reference's CPU batching; unused here."""
    if metric != 0:
        raise NotImplementedError('pseudo-label matching is defined on the image-plane IoU (metric=0), the only mode '
                                  'the reference calls (kitti_dataset_GGA_match.py:423)')
    gt_annos = [info['annos'] for info in gt_infos]
    assert len(gt_annos) == len(dt_annos)
    reserve = copy.deepcopy(gt_infos)
    for anno in gt_annos:
        anno.pop('GGA_in_box_points')           # ragged per-object lists: not an array field
    for anno in gt_annos:                       # DontCare entries trail the list; then keep the 3 trained classes
        n_obj = len([n for n in anno['name'] if n != 'DontCare'])
        for key in anno:
            anno[key] = anno[key][:n_obj]
        keep = drop_arrays_by_name(anno['name'])
        for key in anno:
            anno[key] = anno[key][keep]

    matches = image_box_match([a['bbox'] for a in dt_annos], [a['bbox'] for a in gt_annos], device=device)

    relabelled = []
    for gt, dt, m in zip(gt_annos, dt_annos, matches):
        if len(dt['name']) == 0:
            relabelled.append({key: gt[key][:0] for key in gt})
            continue
        # predicted fields win; GGA side information follows the matched ground truth
        relabelled.append({key: (dt[key] if key in dt else gt[key][m]) for key in gt})

    for sample, anno in zip(reserve, relabelled):
        sample.pop('annos')
        dims, rot = anno['dimensions'], anno['rotation_y']
        for j in range(rot.shape[0]):           # keep the longer horizontal side first: swap and turn by 90 degrees
            if dims[j, 2] > dims[j, 0]:
                dims[j] = dims[j, [2, 1, 0]]
                rot[j] = rot[j] + np.pi / 2.0
        sample['annos'] = anno

    if filename is not None:
        os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
        with open(filename, 'wb') as f:
            pickle.dump(reserve, f)
    return gt_annos
