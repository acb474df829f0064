"""Synthetic KITTI-shaped frames for the GGA train step.

There is no dataset on the build or GPU box, so every test / bench input is a
seeded synthetic frame with the *on-wire format* the reference's collate hands
to ``GGA.forward_train`` (reference: mmdet3d/models/detectors/
mvx_two_stage_gga.py:238-252 and mmdet3d/datasets/pipelines/gga_processing.py):

    points                  [N,4]  f32   (x, y, z, reflectance)
    gt_labels_3d            [n]    i64   0=Pedestrian 1=Cyclist 2=Car
    gt_bboxes_3d            Boxes3D (n x 7, bottom-centre; debug only)
    GGA_boxes_img           [n,4]  f64   2D box (x1, y1, x2, y2) in pixels
    GGA_lidar2img           [n,4,4] f32  per-object projection matrix
    GGA_init_pseudo_labels  [n,7]  f64   (x, y, z, l, w, h, rot)
    GGA_bdry_masks          [n,4]  bool  side touches the image border
    GGA_in_box_points       list of [Ni,4] f64 (x, y, z, 1)
    img_meta['lidar2img']   [4,4]  f32 numpy

Seed = 1234 + 1000*rank + frame_idx (numpy ``default_rng``).
"""
from __future__ import annotations

import numpy as np
import torch

IMG_W, IMG_H = 1242, 375
# (l, w, h) class means, order Pedestrian / Cyclist / Car
# (reference: configs/_base_/models/hv_second_secfpn_kitti.py:40 anchor sizes)
CLASS_DIMS = np.array([[0.8, 0.6, 1.73], [1.76, 0.6, 1.73], [3.9, 1.6, 1.56]])

RANGE_SECOND = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)
RANGE_PP = (0.0, -39.68, -3.0, 69.12, 39.68, 1.0)


class Boxes3D:
    """Minimal stand-in for ``LiDARInstance3DBoxes``: the GGA head only reads
    ``.gravity_center`` and ``.tensor`` (centerpoint_head_gga.py:416-418)."""

    def __init__(self, tensor):
        self.tensor = torch.as_tensor(tensor, dtype=torch.float32).reshape(-1, 7)

    @property
    def gravity_center(self):
        t = self.tensor
        out = t[:, :3].clone()
        out[:, 2] = t[:, 2] + t[:, 5] * 0.5
        return out

    def to(self, device):
        return Boxes3D(self.tensor.to(device))

    def __len__(self):
        return self.tensor.shape[0]


def kitti_lidar2img(rng=None):
    """KITTI-like P2 @ R0_rect @ Tr_velo_to_cam as a [4,4] f32 matrix."""
    P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728],
                   [0.0, 721.5377, 172.854, 0.2163791],
                   [0.0, 0.0, 1.0, 0.002745884],
                   [0.0, 0.0, 0.0, 1.0]])
    R0 = np.eye(4)
    R0[:3, :3] = np.array([[0.9999239, 0.00983776, -0.00744505],
                           [-0.0098698, 0.9999421, -0.00427846],
                           [0.00740253, 0.00435161, 0.9999631]])
    Tr = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03],
                   [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
                   [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01],
                   [0.0, 0.0, 0.0, 1.0]])
    if rng is not None:  # small per-frame calibration jitter
        Tr = Tr.copy()
        Tr[:3, 3] += rng.normal(0.0, 0.01, 3)
    return (P2 @ R0 @ Tr).astype(np.float32)


def box_corners(box):
    """8 corners of (x,y,z_bottom,l,w,h,rot) — same convention as the head's
    decoder (counter-clockwise yaw about z, bottom-centre origin)."""
    x, y, z, l, w, h, r = box
    ox = np.array([-.5, -.5, -.5, -.5, .5, .5, .5, .5]) * l
    oy = np.array([-.5, -.5, .5, .5, -.5, -.5, .5, .5]) * w
    oz = np.array([0., 1., 1., 0., 0., 1., 1., 0.]) * h
    c, s = np.cos(r), np.sin(r)
    return np.stack([x + c * ox - s * oy, y + s * ox + c * oy, z + oz], 1)


def make_frame(frame_idx=0, rank=0, n_points=20000, pc_range=RANGE_SECOND,
               n_obj_range=(4, 20), n_ibp_range=(20, 1500), outside_frac=0.05,
               unlabeled_frac=0.0):
    """One synthetic frame as a dict of the ``forward_train`` keyword inputs."""
    rng = np.random.default_rng(1234 + 1000 * rank + frame_idx)
    x0, y0, z0, x1, y1, z1 = pc_range

    n_obj = int(rng.integers(n_obj_range[0], n_obj_range[1] + 1))
    labels = rng.integers(0, 3, n_obj).astype(np.int64)
    if unlabeled_frac > 0:
        labels[rng.random(n_obj) < unlabeled_frac] = -1
    dims = CLASS_DIMS[np.clip(labels, 0, 2)] * rng.uniform(0.8, 1.2, (n_obj, 3))
    cx = rng.uniform(x0 + 6.0, x1 - 4.0, n_obj)
    cy = rng.uniform(y0 + 8.0, y1 - 8.0, n_obj)
    # keep most objects inside the camera frustum (|y| < 0.8 x) but let a few
    # cross the image border so the boundary masks are exercised
    cy = np.clip(cy, -0.85 * cx, 0.85 * cx)
    cz = rng.uniform(-1.9, -1.3, n_obj)
    rot = rng.uniform(-np.pi, np.pi, n_obj)
    pseudo = np.stack([cx, cy, cz, dims[:, 0], dims[:, 1], dims[:, 2], rot], 1)

    l2i = kitti_lidar2img(rng)
    boxes_img = np.zeros((n_obj, 4), np.float64)
    bdry = np.zeros((n_obj, 4), bool)
    ibp = []
    cluster_pts = []
    l2i_obj = np.repeat(l2i[None], n_obj, 0).copy()
    for j in range(n_obj):
        # tiny per-object perturbation so a slot that picks the wrong matrix
        # is caught by parity tests
        l2i_obj[j, 0, 3] += np.float32(0.01 * j)
        cor = box_corners(pseudo[j])
        q = np.concatenate([cor, np.ones((8, 1))], 1) @ l2i_obj[j].astype(np.float64).T
        d = np.maximum(q[:, 2], 0.1)
        u, v = q[:, 0] / d, q[:, 1] / d
        b = np.array([u.min(), v.min(), u.max(), v.max()]) + rng.uniform(-3, 3, 4)
        clipped = np.array([max(b[0], 0.0), max(b[1], 0.0),
                            min(b[2], IMG_W - 1.0), min(b[3], IMG_H - 1.0)])
        bdry[j] = clipped != b
        boxes_img[j] = clipped
        # in-box points: footprint stretched x1.15 so some fall outside
        ni = int(rng.integers(n_ibp_range[0], n_ibp_range[1] + 1))
        lx = rng.uniform(-0.575, 0.575, ni) * pseudo[j, 3]
        ly = rng.uniform(-0.575, 0.575, ni) * pseudo[j, 4]
        lz = rng.uniform(0.0, 1.0, ni) * pseudo[j, 5]
        c, s = np.cos(rot[j]), np.sin(rot[j])
        p = np.stack([cx[j] + c * lx - s * ly, cy[j] + s * lx + c * ly,
                      cz[j] + lz, np.ones(ni)], 1)
        ibp.append(p)
        cluster_pts.append(p[: min(ni, 200), :3])

    n_out = int(round(n_points * outside_frac))
    cl = np.concatenate(cluster_pts, 0) if cluster_pts else np.zeros((0, 3))
    n_cl = min(len(cl), n_points // 4)
    n_in = n_points - n_out - n_cl
    pts = np.empty((n_points, 4), np.float32)
    pts[:n_in, 0] = rng.uniform(x0, x1, n_in)
    pts[:n_in, 1] = rng.uniform(y0, y1, n_in)
    pts[:n_in, 2] = np.clip(rng.normal(-1.0, 0.6, n_in), z0 + 1e-3, z1 - 1e-3)
    pts[n_in:n_in + n_cl, :3] = cl[:n_cl]
    # points outside the range on every side (exercise the rejection branch)
    o = slice(n_in + n_cl, n_points)
    side = rng.integers(0, 6, n_out)
    ox = rng.uniform(x0, x1, n_out)
    oy = rng.uniform(y0, y1, n_out)
    oz = rng.uniform(z0, z1, n_out)
    ox = np.where(side == 0, x0 - rng.uniform(0.01, 5, n_out), ox)
    ox = np.where(side == 1, x1 + rng.uniform(0.0, 5, n_out), ox)
    oy = np.where(side == 2, y0 - rng.uniform(0.01, 5, n_out), oy)
    oy = np.where(side == 3, y1 + rng.uniform(0.0, 5, n_out), oy)
    oz = np.where(side == 4, z0 - rng.uniform(0.01, 2, n_out), oz)
    oz = np.where(side == 5, z1 + rng.uniform(0.0, 2, n_out), oz)
    pts[o, 0], pts[o, 1], pts[o, 2] = ox, oy, oz
    pts[:, 3] = rng.uniform(0, 1, n_points)
    pts = pts[rng.permutation(n_points)]  # PointShuffle

    return dict(
        points=torch.from_numpy(pts),
        gt_labels_3d=torch.from_numpy(labels),
        gt_bboxes_3d=Boxes3D(pseudo.astype(np.float32)),
        GGA_boxes_img=torch.from_numpy(boxes_img),
        GGA_lidar2img=torch.from_numpy(l2i_obj),
        GGA_init_pseudo_labels=torch.from_numpy(pseudo),
        GGA_bdry_masks=torch.from_numpy(bdry),
        GGA_in_box_points=[torch.from_numpy(p) for p in ibp],
        img_meta=dict(lidar2img=l2i, sample_idx=frame_idx),
    )


BATCH_KEYS = ('points', 'gt_labels_3d', 'gt_bboxes_3d', 'GGA_boxes_img',
              'GGA_lidar2img', 'GGA_init_pseudo_labels', 'GGA_bdry_masks',
              'GGA_in_box_points')


def make_batch(batch_size, start=0, rank=0, device=None, **kw):
    """Collate ``batch_size`` frames into the list-of-per-frame layout the
    reference's ``DataContainer`` collate produces. Tensors that the reference
    moves to the device in ``scatter`` (points, labels, GGA_*) are moved here;
    in-box points stay on the host (the head moves them, head:469-470)."""
    frames = [make_frame(start + i, rank, **kw) for i in range(batch_size)]
    batch = {k: [f[k] for f in frames] for k in BATCH_KEYS}
    batch['img_metas'] = [f['img_meta'] for f in frames]
    if device is not None:
        for k in ('points', 'gt_labels_3d', 'GGA_boxes_img', 'GGA_lidar2img',
                  'GGA_init_pseudo_labels', 'GGA_bdry_masks'):
            batch[k] = [t.to(device) for t in batch[k]]
    return batch


def det_uniform(shape, seed, lo=-0.5, hi=0.5):
    """Platform-independent pseudo-random f32 tensor: an integer hash of the
    flat index (exact in uint64 arithmetic), so golden tests can regenerate
    dense head maps from a seed instead of storing them."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    x = i + np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)  # wraps mod 2^64
    x ^= x >> np.uint64(33)
    x = (x * np.uint64(0xFF51AFD7ED558CCD)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(33)
    x = (x * np.uint64(0xC4CEB9FE1A85EC53)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(33)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24-bit mantissa: exact in f32
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


HEAD_KEYS = (('reg', 2), ('height', 1), ('dim', 3), ('rot', 2), ('heatmap', 1))


def make_head_preds(B, H, W, seed=77, n_tasks=3):
    """Deterministic head outputs ``[{reg,height,dim,rot,heatmap}] * n_tasks``
    shaped like ``CenterHead_GGA.forward`` results (values in a plausible
    range: log-dims ~ U(-1,1), heights ~ -1.5..-0.5, heatmap logits ~ -2.19)."""
    preds = []
    for t in range(n_tasks):
        d = {}
        for j, (k, c) in enumerate(HEAD_KEYS):
            x = det_uniform((B, c, H, W), seed * 100 + t * 10 + j) * 2.0
            if k == 'height':
                x = x * 0.5 - 1.0
            if k == 'heatmap':
                x = x - 2.19
            d[k] = x
        preds.append(d)
    return preds


# ---------------------------------------------------------------------------
# KITTI-info-shaped annotations / detections for the pseudo-label matching step
# ---------------------------------------------------------------------------
KITTI_CLASSES = ['Pedestrian', 'Car', 'Cyclist', 'Van', 'Truck', 'Misc']
PSEUDO_GT_KEYS = ['name', 'truncated', 'occluded', 'alpha', 'bbox', 'dimensions', 'location', 'rotation_y', 'score',
                  'index', 'group_ids', 'difficulty', 'num_points_in_gt', 'GGA_boxes_img', 'GGA_mask_depth',
                  'GGA_mask2d', 'GGA_mask_boundary', 'GGA_bdry_masks', 'GGA_mask_valid', 'GGA_init_pseudo_label',
                  'GGA_num_points_in_box2d']
PSEUDO_DT_KEYS = ['name', 'truncated', 'occluded', 'alpha', 'bbox', 'dimensions', 'location', 'rotation_y', 'score',
                  'sample_idx']


def make_pseudo_case(seed, n_frames, dt_dtype=np.float32):
    """Seeded KITTI-info-shaped ground truths (GGA fields included, DontCare entries last) and
    KITTI-format detections; shared by the golden generator and the tests (inputs are rebuilt
    from the seed, only the expected outputs are stored)."""
    rng = np.random.default_rng(seed)
    infos, dts = [], []
    for f in range(n_frames):
        n_obj = int(rng.integers(0, 9)) if f != 1 else 0
        n_dc = int(rng.integers(0, 3))
        n = n_obj + n_dc
        names = np.array([KITTI_CLASSES[i] for i in rng.integers(0, len(KITTI_CLASSES), n_obj)] + ['DontCare'] * n_dc)
        x1, y1 = rng.uniform(0, 1000, n), rng.uniform(0, 300, n)
        w, h = rng.uniform(10, 250, n), rng.uniform(10, 150, n)
        bbox = np.stack([x1, y1, x1 + w, y1 + h], 1)
        annos = dict(
            name=names, truncated=rng.uniform(0, 1, n), occluded=rng.integers(0, 3, n), alpha=rng.uniform(-3, 3, n),
            bbox=bbox, dimensions=rng.uniform(0.5, 4.5, (n, 3)), location=rng.uniform(-30, 60, (n, 3)),
            rotation_y=rng.uniform(-np.pi, np.pi, n), score=np.zeros(n), index=np.arange(n, dtype=np.int32),
            group_ids=np.arange(n, dtype=np.int32), difficulty=rng.integers(-1, 3, n).astype(np.int32),
            num_points_in_gt=rng.integers(0, 500, n).astype(np.int32), GGA_boxes_img=bbox + rng.uniform(-2, 2, (n, 4)),
            GGA_mask_depth=rng.random(n) < 0.8, GGA_mask2d=rng.random(n) < 0.8, GGA_mask_boundary=rng.random(n) < 0.7,
            GGA_bdry_masks=rng.random((n, 4)) < 0.5, GGA_mask_valid=rng.random(n) < 0.9,
            GGA_init_pseudo_label=rng.uniform(-5, 60, (n, 7)), GGA_num_points_in_box2d=rng.integers(0, 2000, n),
            GGA_in_box_points=[rng.uniform(-1, 1, (int(rng.integers(1, 6)), 4)) for _ in range(n)])
        infos.append(dict(image=dict(image_idx=f, image_shape=np.array([375, 1242], np.int32)),
                          point_cloud=dict(num_features=4), annos=annos))
        trained = [i for i in range(n_obj) if names[i] in KITTI_CLASSES[:3]]
        n_dt = int(rng.integers(0, 7)) if (trained and f != 2) else 0
        # detections: jittered copies of trained-class boxes plus unrelated boxes; a duplicated
        # ground truth (identical IoU for two candidates) exercises the first-maximum rule
        src = rng.choice(trained, n_dt) if n_dt else np.zeros(0, np.int64)
        dbox = bbox[src] + rng.uniform(-15, 15, (n_dt, 4)) * (rng.random((n_dt, 1)) < 0.8)
        far = rng.random(n_dt) < 0.2
        dbox[far] += 5000.0
        dts.append(dict(
            name=np.array([KITTI_CLASSES[i] for i in rng.integers(0, 3, n_dt)]) if n_dt else np.zeros(0, '<U10'),
            truncated=np.zeros(n_dt), occluded=np.zeros(n_dt, np.int64), alpha=rng.uniform(-3, 3, n_dt).astype(dt_dtype),
            bbox=dbox.astype(dt_dtype), dimensions=rng.uniform(0.5, 4.5, (n_dt, 3)).astype(dt_dtype),
            location=rng.uniform(-30, 60, (n_dt, 3)).astype(dt_dtype),
            rotation_y=rng.uniform(-np.pi, np.pi, n_dt).astype(dt_dtype), score=rng.uniform(0.1, 1, n_dt).astype(dt_dtype),
            sample_idx=np.full(n_dt, f, np.int64)))
    if n_frames > 3 and len(infos[3]['annos']['name']) >= 2:      # two identical ground truths in frame 3
        a = infos[3]['annos']
        a['bbox'][1] = a['bbox'][0]
        if a['name'][0] in KITTI_CLASSES[:3]:
            a['name'][1] = a['name'][0]
    return infos, dts


# ---------------------------------------------------------------------------
# Train-pipeline inputs: a ground-truth database and raw frames (before ObjectSample_GGA)
# ---------------------------------------------------------------------------
PIPELINE_CLASSES = ['Pedestrian', 'Cyclist', 'Car']


def make_gt_database(seed, n_per_class=25):
    """-> (db_infos {class: [record]}, points {path: [n,4] f32}); records carry the fields
    DataBaseSampler_GGA reads (gga_processing.py:791-1010)."""
    rng = np.random.default_rng(seed)
    db, pts = {}, {}
    for ci, cname in enumerate(PIPELINE_CLASSES):
        recs = []
        for i in range(n_per_class):
            centre = np.array([rng.uniform(2, 68), rng.uniform(-38, 38), rng.uniform(-1.8, -0.6)])
            n = int(rng.integers(5, 60))
            path = f'db/{cname}_{i}.bin'
            pts[path] = (centre + rng.normal(0, 0.6, (n, 3))).astype(np.float32)
            pts[path] = np.concatenate([pts[path], rng.uniform(0, 1, (n, 1)).astype(np.float32)], 1)
            x1, y1 = rng.uniform(0, 1000), rng.uniform(0, 300)
            recs.append(dict(
                name=cname, path=path, image_idx=int(rng.integers(0, 7000)), gt_idx=i,
                box3d_lidar=np.concatenate([centre, rng.uniform(0.5, 4.0, 3), rng.uniform(-np.pi, np.pi, 1)]).astype(np.float32),
                num_points_in_gt=n, difficulty=np.int64(rng.integers(-1, 3)), group_id=i,
                GGA_init_pseudo_label=np.concatenate([centre + rng.normal(0, 0.2, 3), rng.uniform(0.5, 4.0, 3),
                                                      rng.uniform(-np.pi, np.pi, 1)]),
                GGA_box_img=np.array([x1, y1, x1 + rng.uniform(10, 200), y1 + rng.uniform(10, 120)]),
                GGA_lidar2img=rng.normal(0, 1, (4, 4)), GGA_bdry_mask=rng.random(4) < 0.5,
                GGA_mask2d=np.bool_(rng.random() < 0.9), GGA_mask_depth=np.bool_(rng.random() < 0.9),
                GGA_mask_valid=np.bool_(rng.random() < 0.85), GGA_num_points_in_box2d=np.int64(rng.integers(0, 400)),
                GGA_in_box_points=rng.normal(0, 1, (int(rng.integers(3, 30)), 4))))
        db[cname] = recs
    return db, pts


def make_pipeline_frame(seed, n_points=1200, pc_range=RANGE_SECOND):
    """Raw frame as the loaders hand it to ObjectSample_GGA: plain arrays, objects incl. invalid ones."""
    rng = np.random.default_rng(seed)
    lo, hi = np.array(pc_range[:3]), np.array(pc_range[3:])
    pts = rng.uniform(lo - 3.0, hi + 3.0, (n_points, 3))
    pts[:8] = np.array([[lo[0], 0, -1], [hi[0], 0, -1], [1, lo[1], -1], [1, hi[1], -1], [1, 0, lo[2]], [1, 0, hi[2]],
                        [np.nextafter(np.float32(lo[0]), np.float32(1e9)), 0, -1],
                        [np.nextafter(np.float32(hi[0]), np.float32(-1e9)), 0, -1]])      # on / next to the faces
    pts = np.concatenate([pts, rng.uniform(0, 1, (n_points, 1))], 1).astype(np.float32)
    n = int(rng.integers(2, 9))
    centre = np.stack([rng.uniform(-2, 75, n), rng.uniform(-44, 44, n), rng.uniform(-1.8, -0.6, n)], 1)
    x1, y1 = rng.uniform(0, 1000, n), rng.uniform(0, 300, n)
    return dict(
        points=pts, gt_bboxes_3d=np.concatenate([centre, rng.uniform(0.5, 4, (n, 3)), rng.uniform(-7, 7, (n, 1))], 1).astype(np.float32),
        gt_labels_3d=rng.integers(0, 3, n).astype(np.int64),
        GGA_boxes_img=np.stack([x1, y1, x1 + rng.uniform(10, 200, n), y1 + rng.uniform(10, 120, n)], 1),
        GGA_lidar2img=rng.normal(0, 1, (n, 4, 4)),
        GGA_init_pseudo_labels=np.concatenate([centre + rng.normal(0, 0.2, (n, 3)), rng.uniform(0.5, 4, (n, 3)),
                                               rng.uniform(-np.pi, np.pi, (n, 1))], 1),
        GGA_mask_valid=rng.random(n) < 0.8, GGA_bdry_masks=rng.random((n, 4)) < 0.5,
        GGA_difficulty=rng.integers(-1, 3, n), GGA_num_points_in_box2d=rng.integers(0, 60, n),
        GGA_in_box_points=[rng.normal(0, 1, (int(rng.integers(3, 30)), 4)) for _ in range(n)])


# ---------------------------------------------------------------------------
# Offline label generation inputs (utils_gga.py primitives)
# ---------------------------------------------------------------------------
KITTI_CALIB = dict(          # a KITTI-shaped calibration (4x4, float64)
    P2=np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
                 [0.0, 0.0, 1.0, 0.002745884], [0.0, 0.0, 0.0, 1.0]]),
    R0_rect=np.array([[0.9999239, 0.00983776, -0.00744505, 0.0], [-0.0098698, 0.9999421, -0.00427846, 0.0],
                      [0.00740253, 0.00435161, 0.9999631, 0.0], [0.0, 0.0, 0.0, 1.0]]),
    Tr_velo_to_cam=np.array([[0.007533745, -0.9999714, -0.000616602, -0.004069766],
                             [0.01480249, 0.0007280733, -0.9998902, -0.07631618],
                             [0.9998621, 0.00752379, 0.01480755, -0.2717806], [0.0, 0.0, 0.0, 1.0]]))


def make_region_grow_case(seed, n=700):
    """Camera-frame homogeneous points [n,4] f64: a few object-like clusters over background,
    a search mask and an origin mask (search points inside a window around the first clusters)."""
    rng = np.random.default_rng(seed)
    k = int(rng.integers(3, 6))
    centres = np.stack([rng.uniform(-8, 8, k), rng.uniform(0.5, 1.5, k), rng.uniform(6, 30, k)], 1)
    per = n // (2 * k)
    pts = [c + rng.normal(0, [0.5, 0.35, 0.6], (per, 3)) for c in centres]
    pts.append(np.stack([rng.uniform(-12, 12, n - per * k), rng.uniform(-1, 2, n - per * k), rng.uniform(2, 40, n - per * k)], 1))
    pc = np.concatenate(pts)[rng.permutation(n)]
    pc = np.concatenate([pc, np.ones((n, 1))], 1)
    pc[5] = pc[4]                                            # an exact duplicate: first-minimum argmin matters
    mask_search = (rng.random(n) < 0.85).astype(np.float64)
    win = (np.abs(pc[:, 0] - centres[0, 0]) < 2.2) & (np.abs(pc[:, 2] - centres[0, 2]) < 2.5)
    win |= (np.abs(pc[:, 0] - centres[1, 0]) < 0.8) & (np.abs(pc[:, 2] - centres[1, 2]) < 0.9)
    mask_origin = mask_search * win
    return pc, mask_search, mask_origin


def make_ground_case(seed, n=3000):
    """Camera-frame points [n,3] f64 (y down): a slightly tilted ground plane, objects, clutter."""
    rng = np.random.default_rng(seed)
    ng = int(n * 0.55)
    x, z = rng.uniform(-20, 20, ng), rng.uniform(3, 60, ng)
    ground = np.stack([x, 1.65 + 0.01 * x - 0.004 * z + rng.normal(0, 0.03, ng), z], 1)
    no = n - ng
    obj = np.stack([rng.uniform(-15, 15, no), rng.uniform(-1.5, 1.5, no), rng.uniform(4, 50, no)], 1)
    return np.concatenate([ground, obj])[rng.permutation(n)]


def make_frustum_case(seed, n=2500):
    """LiDAR-frame points [n,4] f64 (homogeneous) in front of the car + a few 2D boxes (x1,y1,x2,y2)."""
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(0, 60, n), rng.uniform(-25, 25, n), rng.uniform(-2.5, 1.0, n), np.ones(n)], 1)
    boxes = np.array([[300.0, 120.0, 520.0, 260.0], [0.0, 0.0, 1241.0, 374.0], [900.0, 150.0, 1241.0, 374.0],
                      [600.5, 170.25, 640.75, 200.5], [-1.0, -1.0, -1.0, -1.0]])
    return pts, boxes


def make_rga_scene(seed, n_ground=1200, n_clutter=500):
    """A KITTI-shaped frame for the offline label generator: velodyne points [N,4] f32 (ground plane,
    a few objects standing on it, clutter), KITTI annotations in camera coordinates (DontCare
    entries last), calibration and image shape."""
    rng = np.random.default_rng(seed)
    calib = {k: v.copy() for k, v in KITTI_CALIB.items()}
    l2c = calib['R0_rect'] @ calib['Tr_velo_to_cam']
    gz = -1.72
    gx, gy = rng.uniform(3, 60, n_ground), rng.uniform(-25, 25, n_ground)
    pts = [np.stack([gx, gy, gz + 0.002 * gx + rng.normal(0, 0.02, n_ground)], 1)]
    sizes = {'Car': (3.9, 1.56, 1.6), 'Pedestrian': (0.8, 1.73, 0.6), 'Cyclist': (1.76, 1.73, 0.6)}     # l, h, w
    names, loc, dims, rot, npts = [], [], [], [], []
    n_obj = int(rng.integers(3, 6))
    for i in range(n_obj):
        cls = ['Car', 'Pedestrian', 'Cyclist'][int(rng.integers(0, 3))] if i else 'Car'
        l, h, w = [s * rng.uniform(0.9, 1.1) for s in sizes[cls]]
        cx, cy = rng.uniform(8, 35), rng.uniform(-8, 8)
        if i == 1:
            cx, cy = 10.0, 8.4 * (1 if seed % 2 else -1)                 # cut by the image border -> truncated-object branch
        yaw = rng.uniform(-np.pi, np.pi)
        n = int(rng.integers(60, 220)) if i != 2 else 0                 # one object without points
        u = np.stack([rng.uniform(-l / 2, l / 2, n), rng.uniform(-w / 2, w / 2, n), rng.uniform(0.15, h, n)], 1)
        c, s = np.cos(yaw), np.sin(yaw)
        pts.append(np.stack([cx + c * u[:, 0] - s * u[:, 1], cy + s * u[:, 0] + c * u[:, 1], gz + u[:, 2]], 1))
        cam = l2c @ np.array([cx, cy, gz, 1.0])
        names.append(cls); loc.append(cam[:3]); dims.append([l, h, w]); rot.append(-yaw - np.pi / 2); npts.append(n)
    pts.append(np.stack([rng.uniform(3, 60, n_clutter), rng.uniform(-25, 25, n_clutter), rng.uniform(gz + 0.3, 1.5, n_clutter)], 1))
    pts.append(np.stack([rng.uniform(-20, 0, 80), rng.uniform(-25, 25, 80), rng.uniform(gz, 1.0, 80)], 1))    # behind the camera
    xyz = np.concatenate(pts)
    xyz = xyz[rng.permutation(len(xyz))]
    points_v = np.concatenate([xyz, rng.uniform(0, 1, (len(xyz), 1))], 1).astype(np.float32)
    n_dc = int(rng.integers(1, 3))
    n = n_obj + n_dc
    annos = dict(
        name=np.array(names + ['DontCare'] * n_dc), truncated=np.zeros(n), occluded=np.zeros(n, np.int64),
        alpha=np.zeros(n), bbox=np.zeros((n, 4)),
        dimensions=np.concatenate([np.array(dims), -np.ones((n_dc, 3))]),
        location=np.concatenate([np.array(loc), -1000 * np.ones((n_dc, 3))]),
        rotation_y=np.concatenate([np.array(rot), -10 * np.ones(n_dc)]), score=np.zeros(n),
        index=np.arange(n, dtype=np.int32), group_ids=np.arange(n, dtype=np.int32),
        difficulty=np.zeros(n, np.int32), num_points_in_gt=np.array(npts + [-1] * n_dc, dtype=np.int32))
    return points_v, calib, annos, (375, 1242)


# ----------------------------------------------------------------------------- mono3d (PGD retraining, BASELINE config #5)
MONO_IMG = (375, 1242)              # KITTI image, padded to a multiple of 32 by the pipeline ('Pad', size_divisor=32)
MONO_CAM2IMG = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]],
                        np.float32)
_MONO_DIMS = np.array([[0.8, 1.73, 0.6], [1.76, 1.73, 0.6], [3.9, 1.56, 1.6]], np.float32)      # (l, h, w) per class


def make_mono_batch(batch_size, start=0, rank=0, device=None, img_hw=MONO_IMG, n_obj_range=(3, 9)):
    """KITTI-mono3d-shaped training batch as configs/gga/gga_pdg.py's pipeline hands it to
    ``FCOSMono3D.forward_train``: normalised image tensor [B, 3, H', W'] (H', W' padded to multiples of 32),
    and per image ``gt_bboxes`` [n,4], ``gt_labels`` [n], ``gt_bboxes_3d`` [n,7] (camera boxes: x, y, z, l, h, w, yaw),
    ``gt_labels_3d``, ``centers2d`` [n,2], ``depths`` [n], ``img_metas`` with ``cam2img`` and the camera box type.
    Seeded by ``1234 + 1000 * rank + frame`` like the LiDAR frames."""
    from .box3d import CameraInstance3DBoxes
    H, W = img_hw
    Hp, Wp = -(-H // 32) * 32, -(-W // 32) * 32
    out = dict(img=[], img_metas=[], gt_bboxes=[], gt_labels=[], gt_bboxes_3d=[], gt_labels_3d=[], centers2d=[], depths=[])
    fx, cx, cy = float(MONO_CAM2IMG[0, 0]), float(MONO_CAM2IMG[0, 2]), float(MONO_CAM2IMG[1, 2])
    for f in range(start, start + batch_size):
        rng = np.random.default_rng(1234 + 1000 * rank + f)
        img = np.zeros((3, Hp, Wp), np.float32)
        img[:, :H, :W] = rng.normal(0.0, 40.0, (3, H, W)).astype(np.float32)        # mean-subtracted BGR, std 1 (img_norm_cfg)
        n = int(rng.integers(*n_obj_range))
        labels = rng.integers(0, 3, n)
        depth = rng.uniform(6.0, 55.0, n).astype(np.float32)
        c2d = np.stack([rng.uniform(40, W - 40, n), rng.uniform(150, H - 30, n)], 1).astype(np.float32)
        dims = _MONO_DIMS[labels] * rng.uniform(0.9, 1.1, (n, 3)).astype(np.float32)
        xyz = np.stack([(c2d[:, 0] - cx) * depth / fx, (c2d[:, 1] - cy) * depth / fx, depth], 1).astype(np.float32)
        yaw = rng.uniform(-np.pi, np.pi, (n, 1)).astype(np.float32)
        half = np.stack([dims[:, 0] + dims[:, 2], dims[:, 1]], 1) * fx / depth[:, None] * 0.5       # projected half extent
        boxes = np.concatenate([c2d - half, c2d + half], 1)
        boxes[:, [0, 2]] = boxes[:, [0, 2]].clip(0, W - 1)
        boxes[:, [1, 3]] = boxes[:, [1, 3]].clip(0, H - 1)
        t = lambda a, dt=torch.float32: torch.as_tensor(a, dtype=dt, device=device)
        out['img'].append(t(img))
        out['gt_bboxes'].append(t(boxes))
        out['gt_labels'].append(t(labels, torch.int64))
        out['gt_bboxes_3d'].append(t(np.concatenate([xyz, dims, yaw], 1)))
        out['gt_labels_3d'].append(t(labels, torch.int64))
        out['centers2d'].append(t(c2d))
        out['depths'].append(t(depth))
        out['img_metas'].append(dict(cam2img=MONO_CAM2IMG.tolist(), box_type_3d=CameraInstance3DBoxes, img_shape=(H, W, 3),
                                     pad_shape=(Hp, Wp, 3), ori_shape=(H, W, 3), scale_factor=1.0, flip=False))
    out['img'] = torch.stack(out['img'])
    return out


MONO_BATCH_KEYS = ('img', 'img_metas', 'gt_bboxes', 'gt_labels', 'gt_bboxes_3d', 'gt_labels_3d', 'centers2d', 'depths')


def damp_random_backbone(model, scale=0.2):
    """Random-init stand-in for the model-zoo checkpoint configs/gga/gga_pdg.py starts from (not available offline): its
    ResNet-101 runs with BatchNorm frozen in evaluation mode, i.e. as the identity on fresh statistics, and 33 un-normalised
    residual blocks at Kaiming scale double the activation variance per block. Scales the last norm of every bottleneck so
    synthetic runs stay finite; every kernel launched and every trainable parameter's gradient path stays (non-zero)."""
    import torch
    with torch.no_grad():
        for m in model.backbone.modules():
            if hasattr(m, 'bn3'):
                m.bn3.weight.fill_(scale)
    return model


INDOOR_CLASSES = ('bed', 'table', 'sofa', 'chair', 'toilet', 'desk', 'dresser', 'night_stand', 'bookshelf', 'bathtub')
INDOOR_BATCH_KEYS = ('points', 'gt_bboxes_3d', 'gt_labels_3d', 'img_metas')


def make_indoor_batch(batch_size, start=0, rank=0, device=None, n_points=50000, n_obj_range=(4, 12)):
    """SUN RGB-D-shaped training batch as configs/fcaf3d/fcaf3d_8x2_sunrgbd-3d-10class.py's pipeline hands it to
    ``MinkSingleStage3DDetector.forward_train`` (BASELINE config 4: 50 k points, 10 classes): per scene ``points`` [n, 6]
    (x, y, z in metres, depth coordinates, + r, g, b in [0, 1]), ``gt_bboxes_3d`` (DepthInstance3DBoxes with yaw, given - as
    the dataset gives them - by gravity centre), ``gt_labels_3d``. A room of about 5 x 5 x 2.7 m seen from inside: points on
    the floor, two walls and the surfaces of the objects. Seeded by ``1234 + 1000 * rank + scene`` like the LiDAR frames."""
    import torch
    from .fcaf3d import DepthInstance3DBoxes
    out = dict(points=[], gt_bboxes_3d=[], gt_labels_3d=[], img_metas=[])
    for f in range(start, start + batch_size):
        rng = np.random.default_rng(1234 + 1000 * rank + f)
        n_obj = int(rng.integers(n_obj_range[0], n_obj_range[1] + 1))
        size = rng.uniform([0.4, 0.4, 0.4], [2.0, 1.6, 1.2], (n_obj, 3))
        ctr = np.concatenate([rng.uniform(-2.0, 2.0, (n_obj, 2)), size[:, 2:] / 2 + rng.uniform(0, 0.3, (n_obj, 1))], 1)
        yaw = rng.uniform(-np.pi / 2, np.pi / 2, (n_obj, 1))
        labels = rng.integers(0, len(INDOOR_CLASSES), n_obj)
        n_bg = n_points // 2
        n_per = (n_points - n_bg) // n_obj
        pts = []
        floor = np.concatenate([rng.uniform(-2.5, 2.5, (n_bg // 2, 2)), rng.normal(0, 0.004, (n_bg // 2, 1))], 1)
        wall1 = np.stack([rng.uniform(-2.5, 2.5, n_bg // 4), np.full(n_bg // 4, 2.5) + rng.normal(0, 0.004, n_bg // 4), rng.uniform(0, 2.7, n_bg // 4)], 1)
        n_w2 = n_bg - n_bg // 2 - n_bg // 4
        wall2 = np.stack([np.full(n_w2, -2.5) + rng.normal(0, 0.004, n_w2), rng.uniform(-2.5, 2.5, n_w2), rng.uniform(0, 2.7, n_w2)], 1)
        pts += [floor, wall1, wall2]
        for i in range(n_obj):
            n_i = n_per if i < n_obj - 1 else n_points - n_bg - n_per * (n_obj - 1)
            u = rng.uniform(-0.5, 0.5, (n_i, 3))
            face = rng.integers(0, 3, n_i)
            u[np.arange(n_i), face] = np.where(rng.random(n_i) < 0.5, -0.5, 0.5)       # on the surface of the unit box
            local = u * size[i]
            c, s = np.cos(yaw[i, 0]), np.sin(yaw[i, 0])
            xy = np.stack([local[:, 0] * c - local[:, 1] * s, local[:, 0] * s + local[:, 1] * c], 1)
            pts.append(np.concatenate([xy + ctr[i, :2], local[:, 2:] + ctr[i, 2]], 1))
        xyz = np.concatenate(pts).astype(np.float32)
        xyz = xyz[rng.permutation(len(xyz))]
        rgb = rng.uniform(0, 1, (len(xyz), 3)).astype(np.float32)
        p = torch.from_numpy(np.concatenate([xyz, rgb], 1))
        out['points'].append(p.to(device) if device is not None else p)
        out['gt_bboxes_3d'].append(DepthInstance3DBoxes(torch.from_numpy(np.concatenate([ctr, size, yaw], 1).astype(np.float32)),
                                                        box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0.5)))
        out['gt_labels_3d'].append(torch.from_numpy(labels.astype(np.int64)).to(device) if device is not None
                                   else torch.from_numpy(labels.astype(np.int64)))
        out['img_metas'].append(dict(box_type_3d=DepthInstance3DBoxes, sample_idx=f))
    return out


# ---------------------------------------------------------------------------
# An on-disk KITTI tree in the layout the GGA train job reads (loader-fed bench / tests)
# ---------------------------------------------------------------------------
def _frame_objects(rng, n_obj, pc_range, n_ibp_range, calib):
    """``n_obj`` labelled objects of one frame: LiDAR pseudo boxes, their camera-frame annotation, 2D boxes from the projected
    corners, boundary flags, in-box points ([Ni,4] f64, x y z 1) and the first 200 of them as scene points."""
    from .datasets import lidar_boxes_to_camera
    x0, y0, _, x1, y1, _ = pc_range
    labels = rng.integers(0, 3, n_obj)
    dims = CLASS_DIMS[labels] * rng.uniform(0.8, 1.2, (n_obj, 3))
    cx = rng.uniform(x0 + 6.0, x1 - 4.0, n_obj)
    cy = np.clip(rng.uniform(y0 + 8.0, y1 - 8.0, n_obj), -0.85 * cx, 0.85 * cx)
    cz = rng.uniform(-1.9, -1.3, n_obj)
    rot = rng.uniform(-np.pi, np.pi, n_obj)
    pseudo = np.stack([cx, cy, cz, dims[:, 0], dims[:, 1], dims[:, 2], rot], 1)
    l2i = (calib['P2'] @ calib['R0_rect'] @ calib['Tr_velo_to_cam'])
    boxes_img, bdry, ibp, scene = np.zeros((n_obj, 4)), np.zeros((n_obj, 4), bool), [], []
    for j in range(n_obj):
        q = np.concatenate([box_corners(pseudo[j]), np.ones((8, 1))], 1) @ l2i.T
        d = np.maximum(q[:, 2], 0.1)
        u, v = q[:, 0] / d, q[:, 1] / d
        b = np.array([u.min(), v.min(), u.max(), v.max()]) + rng.uniform(-3, 3, 4)
        clipped = np.array([max(b[0], 0.0), max(b[1], 0.0), min(b[2], IMG_W - 1.0), min(b[3], IMG_H - 1.0)])
        bdry[j], boxes_img[j] = clipped != b, clipped
        ni = int(rng.integers(n_ibp_range[0], n_ibp_range[1] + 1))
        lx, ly = rng.uniform(-0.575, 0.575, ni) * pseudo[j, 3], rng.uniform(-0.575, 0.575, ni) * pseudo[j, 4]
        lz = rng.uniform(0.0, 1.0, ni) * pseudo[j, 5]
        c, s = np.cos(rot[j]), np.sin(rot[j])
        p = np.stack([cx[j] + c * lx - s * ly, cy[j] + s * lx + c * ly, cz[j] + lz, np.ones(ni)], 1)
        ibp.append(p)
        scene.append(p[:min(ni, 200), :3])
    cam = lidar_boxes_to_camera(torch.from_numpy(pseudo.astype(np.float32)), (calib['R0_rect'] @ calib['Tr_velo_to_cam']).astype(np.float32)).numpy()
    return labels, pseudo, cam, boxes_img, bdry, ibp, scene


def write_kitti_tree(root, n_frames, n_points=20000, pc_range=RANGE_PP, n_obj_range=(4, 12), n_ibp_range=(20, 1500),
                     db_per_class=300, seed=4100, pts_prefix='velodyne_reduced', info_name='kitti_infos_trainval_GGA.pkl',
                     db_name='kitti_dbinfos_train_GGA.pkl'):
    """Writes what ``configs/gga/gga_kitti_config.py``'s ``data.train`` reads (reference: mmdet3d/datasets/
    kitti_dataset_GGA_train.py:100-329, tools/data_converter/kitti_converter_gga.py:214-517, create_gt_database_gga.py:236-420)
    for ``n_frames`` synthetic frames of SURVEY 8(d)'s shape under ``root``:

        training/<pts_prefix>/%06d.bin          [n_points,4] f32 scans
        <info_name>                             the info list (calib, camera-frame annos, all GGA_* fields, in-box points)
        kitti_gt_database_GGA/*.bin + <db_name> ``db_per_class`` database objects per class (absolute-coordinate points)

    -> (info path, database-info path). Deterministic in ``seed``; a tree that already holds ``n_frames`` scans and both
    pickles written with the same arguments is left alone (a stamp file records them)."""
    import json
    import os
    import pickle
    stamp = dict(n_frames=n_frames, n_points=n_points, pc_range=list(pc_range), n_obj_range=list(n_obj_range),
                 n_ibp_range=list(n_ibp_range), db_per_class=db_per_class, seed=seed, pts_prefix=pts_prefix, v=2)
    info_path, db_path, stamp_path = os.path.join(root, info_name), os.path.join(root, db_name), os.path.join(root, 'synthetic_tree.json')
    if os.path.exists(stamp_path) and json.load(open(stamp_path)) == stamp and os.path.exists(info_path) and os.path.exists(db_path):
        return info_path, db_path
    velo = os.path.join(root, 'training', pts_prefix)
    db_dir = os.path.join(root, 'kitti_gt_database_GGA')
    os.makedirs(velo, exist_ok=True), os.makedirs(db_dir, exist_ok=True)
    names = np.array(PIPELINE_CLASSES)
    x0, y0, z0, x1, y1, z1 = pc_range
    calib = {k: v.copy() for k, v in KITTI_CALIB.items()}
    infos = []
    for i in range(n_frames):
        rng = np.random.default_rng(seed + i)
        n_obj = int(rng.integers(n_obj_range[0], n_obj_range[1] + 1))
        labels, pseudo, cam, boxes_img, bdry, ibp, scene = _frame_objects(rng, n_obj, pc_range, n_ibp_range, calib)
        cl = np.concatenate(scene, 0)
        n_cl = min(len(cl), n_points // 4)
        n_out = int(round(0.05 * n_points))
        n_in = n_points - n_cl - n_out
        pts = np.empty((n_points, 4), np.float32)
        pts[:n_in, 0], pts[:n_in, 1] = rng.uniform(x0, x1, n_in), rng.uniform(y0, y1, n_in)
        pts[:n_in, 2] = np.clip(rng.normal(-1.0, 0.6, n_in), z0 + 1e-3, z1 - 1e-3)
        pts[n_in:n_in + n_cl, :3] = cl[:n_cl]
        far = rng.uniform([x0 - 5, y0 - 5, z0 - 2], [x1 + 5, y1 + 5, z1 + 2], (n_out, 3))          # 5 % around the range: the
        pts[n_in + n_cl:, :3] = far                                                                # filter has something to reject
        pts[:, 3] = rng.uniform(0, 1, n_points)
        pts[rng.permutation(n_points)].tofile(os.path.join(velo, f'{i:06d}.bin'))
        # one DontCare region per frame (dropped by the loader), objects of unknown difficulty now and then
        difficulty = rng.integers(0, 3, n_obj + 1).astype(np.int32)
        difficulty[rng.random(n_obj + 1) < 0.03] = -1
        name = np.concatenate([names[labels], ['DontCare']])
        pad = lambda a, fill: np.concatenate([a, np.full((1,) + a.shape[1:], fill, a.dtype)], 0)
        annos = dict(
            name=name, truncated=np.zeros(n_obj + 1), occluded=np.zeros(n_obj + 1, np.int64), alpha=pad(-cam[:, 6].astype(np.float64), -10.0),
            bbox=pad(boxes_img, 0.0), dimensions=pad(cam[:, 3:6].astype(np.float64), -1.0), location=pad(cam[:, :3].astype(np.float64), -1000.0),
            rotation_y=pad(cam[:, 6].astype(np.float64), -10.0), score=np.zeros(n_obj + 1), index=np.concatenate([np.arange(n_obj), [-1]]).astype(np.int32),
            group_ids=np.arange(n_obj + 1, dtype=np.int32), difficulty=difficulty,
            num_points_in_gt=pad(np.array([len(p) for p in ibp], np.int32), -1),
            GGA_boxes_img=pad(boxes_img, 0.0), GGA_mask_depth=pad(np.ones(n_obj, bool), False), GGA_mask2d=pad(np.ones(n_obj, bool), False),
            GGA_mask_boundary=pad(bdry.any(1), False), GGA_bdry_masks=pad(bdry, False),
            GGA_mask_valid=pad(rng.random(n_obj) < 0.95, False), GGA_in_box_points=ibp + [np.zeros((0, 4))],
            GGA_init_pseudo_label=pad(pseudo, 0.0), GGA_num_points_in_box2d=pad(np.array([float(len(p)) for p in ibp]), 0.0))
        infos.append(dict(point_cloud=dict(velodyne_path=f'training/{pts_prefix}/{i:06d}.bin', num_features=4),
                          image=dict(image_idx=i, image_shape=np.array([IMG_H, IMG_W], np.int32), image_path=f'training/image_2/{i:06d}.png'),
                          calib=calib, annos=annos))
    with open(info_path, 'wb') as f:
        pickle.dump(infos, f)
    # the database: objects drawn like the frames' own, points = the object's frustum points in absolute coordinates
    l2i32 = (calib['P2'].astype(np.float32) @ calib['R0_rect'].astype(np.float32) @ calib['Tr_velo_to_cam'].astype(np.float32))
    rng = np.random.default_rng(seed - 1)
    db = {c: [] for c in PIPELINE_CLASSES}
    while min(len(v) for v in db.values()) < db_per_class:
        labels, pseudo, cam, boxes_img, bdry, ibp, _ = _frame_objects(rng, 12, pc_range, n_ibp_range, calib)
        for j in range(12):
            cname = PIPELINE_CLASSES[labels[j]]
            if len(db[cname]) >= db_per_class:
                continue
            k = len(db[cname])
            path = os.path.join('kitti_gt_database_GGA', f'{k}_{cname}_{j}.bin')
            obj = np.concatenate([ibp[j][:, :3], rng.uniform(0, 1, (len(ibp[j]), 1))], 1).astype(np.float32)
            obj.tofile(os.path.join(root, path))
            db[cname].append(dict(
                name=cname, path=path, image_idx=k, gt_idx=j, box3d_lidar=pseudo[j].astype(np.float32), num_points_in_gt=len(obj),
                difficulty=np.int32(rng.integers(0, 3)), group_id=len(db[cname]), GGA_gt_box=boxes_img[j].astype(np.float32),
                GGA_box_img=boxes_img[j], GGA_mask_depth=np.bool_(True), GGA_mask2d=np.bool_(True), GGA_mask_valid=np.bool_(rng.random() < 0.95),
                GGA_mask_boundary=np.bool_(bdry[j].any()), GGA_bdry_mask=bdry[j], GGA_in_box_points=ibp[j], GGA_init_pseudo_label=pseudo[j],
                GGA_num_points_in_box2d=np.float64(len(obj)), GGA_lidar2img=l2i32))
    with open(db_path, 'wb') as f:
        pickle.dump(db, f)
    json.dump(stamp, open(stamp_path, 'w'))
    return info_path, db_path
