"""Python face of the CPU oracle. TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``gga_amd/``) never does.

* thin ctypes wrappers over ``oracle/libgga_oracle.so`` (gga_oracle.c), and
* the host-side glue of the reference's ``CenterHead_GGA.get_targets`` /
  ``loss`` (mmdet3d/models/dense_heads/centerpoint_head_gga.py:343-723),
  restated as plain loops over frames / tasks / objects.

Parity is pinned by tests/test_oracle.py against tests/golden/*.npz (outputs of
the imported reference, see tools_dev/make_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
f64p = np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS')
i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')
i64p = np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')


def build(force=False):
    so = os.path.join(_HERE, 'libgga_oracle.so')
    src = os.path.join(_HERE, 'gga_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libgga_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.gga_oracle_grid_size.argtypes = [f32p, f32p, i32p]
        L.gga_oracle_hard_voxelize.restype = C.c_int64
        L.gga_oracle_hard_voxelize.argtypes = [f32p, C.c_int64, C.c_int, f32p, f32p, C.c_int,
                                               C.c_int, f32p, i32p, i32p]
        L.gga_oracle_voxel_mean.argtypes = [f32p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int, f32p]
        L.gga_oracle_pfn_decorate.argtypes = [f32p, i32p, i32p, C.c_int64, C.c_int] + [C.c_float] * 6 + [f32p]
        L.gga_oracle_pfn_layer.argtypes = [f32p, C.c_int64, C.c_int, C.c_int, f32p, C.c_int, f32p, f32p,
                                           C.c_float, f32p, f32p, f32p]
        L.gga_oracle_pillar_scatter.argtypes = [f32p, i32p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                                C.c_int, f32p]
        L.gga_oracle_gaussian_radius.restype = C.c_double
        L.gga_oracle_gaussian_radius.argtypes = [C.c_double] * 3
        L.gga_oracle_draw_gaussian.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.gga_oracle_focal_loss.restype = C.c_float
        L.gga_oracle_focal_loss.argtypes = [f32p, f32p, C.c_int64, C.c_float, C.c_float,
                                            C.c_void_p, C.POINTER(C.c_double)]
        L.gga_oracle_gather_pred.argtypes = [f32p, f32p, f32p, f32p, i64p, C.c_int, C.c_int,
                                             C.c_int, C.c_int, f32p]
        L.gga_oracle_box_project.argtypes = [f32p, i64p, f32p, C.c_int64, C.c_int] + [C.c_float] * 5 + [f32p] * 4
        L.gga_oracle_pal_object.argtypes = [f64p, C.c_int64, C.c_int, f32p, f32p]
        L.gga_oracle_l1_loss.restype = C.c_float
        L.gga_oracle_l1_loss.argtypes = [f32p, f32p, f32p, C.c_int64, C.c_float, C.c_float]
        L.gga_oracle_rotated_iou.restype = C.c_float
        L.gga_oracle_rotated_iou.argtypes = [f32p, f32p, C.c_int]
        L.gga_oracle_nms_rotated_sorted.restype = C.c_int
        L.gga_oracle_nms_rotated_sorted.argtypes = [f32p, C.c_int, C.c_float, i64p]
        L.gga_oracle_points_in_boxes.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, i32p]
        L.gga_oracle_image_box_overlap.argtypes = [f64p, C.c_int, f64p, C.c_int, C.c_int, f64p]
        u8p = np.ctypeslib.ndpointer(np.uint8, flags='C_CONTIGUOUS')
        L.gga_oracle_region_grow.argtypes = [f64p, C.c_int64, C.c_int, u8p, u8p, C.c_double, C.c_double, C.c_int, u8p]
        L.gga_oracle_points_in_polyhedra.argtypes = [f64p, C.c_int64, C.c_int, f64p, f64p, C.c_int, C.c_int, u8p]
        L.gga_oracle_plane_inliers.restype = C.c_int64
        L.gga_oracle_plane_inliers.argtypes = [f64p, C.c_int64, C.c_int, f64p, C.c_double, u8p]
        L.gga_oracle_points_prepare.restype = C.c_int64
        L.gga_oracle_points_prepare.argtypes = [f32p, C.c_int64, f32p, C.c_int64, f64p, C.c_int64, C.c_int, C.c_double, f32p, f32p]
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


# ---------------------------------------------------------------------------
# thin wrappers
# ---------------------------------------------------------------------------
def grid_size(voxel_size, pc_range):
    g = np.zeros(3, np.int32)
    lib().gga_oracle_grid_size(_f32(voxel_size), _f32(pc_range), g)
    return g


def hard_voxelize(points, voxel_size, pc_range, max_points, max_voxels):
    """-> voxels [M,P,C] f32, coors [M,3] i32 (z,y,x), num_points [M] i32."""
    points = _f32(points)
    n, ndim = points.shape
    voxels = np.zeros((max_voxels, max_points, ndim), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    npv = np.zeros(max_voxels, np.int32)
    m = lib().gga_oracle_hard_voxelize(points, n, ndim, _f32(voxel_size), _f32(pc_range),
                                       max_points, max_voxels, voxels, coors, npv)
    return voxels[:m], coors[:m], npv[:m]


def voxelize_batch(points_list, voxel_size, pc_range, max_points, max_voxels):
    """MVXTwoStageDetector_GGA.voxelize (mvx_two_stage_gga.py:211-236)."""
    V, Cc, N = [], [], []
    for b, p in enumerate(points_list):
        v, c, n = hard_voxelize(p, voxel_size, pc_range, max_points, max_voxels)
        V.append(v), N.append(n)
        Cc.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    return np.concatenate(V), np.concatenate(N), np.concatenate(Cc)


def voxel_mean(voxels, num_points, num_features=4):
    voxels = _f32(voxels)
    m, P, ndim = voxels.shape
    out = np.zeros((m, num_features), np.float32)
    lib().gga_oracle_voxel_mean(voxels, np.ascontiguousarray(num_points, np.int32), m, P, ndim,
                                num_features, out)
    return out


def pfn_forward(voxels, num_points, coors4, voxel_size, pc_range, W, gamma, beta, eps=1e-3):
    """PillarFeatureNet (single PFNLayer, legacy=True, training-mode BN)."""
    voxels = _f32(voxels)
    m, P, _ = voxels.shape
    vx, vy, vz = (float(v) for v in voxel_size)
    xo, yo, zo = vx / 2 + pc_range[0], vy / 2 + pc_range[1], vz / 2 + pc_range[2]
    feats = np.zeros((m, P, 10), np.float32)
    lib().gga_oracle_pfn_decorate(voxels, np.ascontiguousarray(num_points, np.int32),
                                  np.ascontiguousarray(coors4, np.int32), m, P, vx, vy, vz, xo, yo, zo, feats)
    Cout = W.shape[0]
    out = np.zeros((m, Cout), np.float32)
    mean = np.zeros(Cout, np.float32)
    var = np.zeros(Cout, np.float32)
    lib().gga_oracle_pfn_layer(feats, m, P, 10, _f32(W), Cout, _f32(gamma), _f32(beta), eps, out, mean, var)
    return out, feats, mean, var


def pillar_scatter(feats, coors4, batch_size, ny, nx):
    feats = _f32(feats)
    m, Cc = feats.shape
    canvas = np.zeros((batch_size, Cc, ny, nx), np.float32)
    lib().gga_oracle_pillar_scatter(feats, np.ascontiguousarray(coors4, np.int32), m, Cc,
                                    batch_size, ny, nx, canvas)
    return canvas


def gaussian_radius(height, width, min_overlap):
    return lib().gga_oracle_gaussian_radius(float(height), float(width), float(min_overlap))


def draw_gaussian(heatmap, cx, cy, radius):
    H, W = heatmap.shape
    lib().gga_oracle_draw_gaussian(heatmap, H, W, int(cx), int(cy), int(radius))
    return heatmap


def focal_loss(logits, target, alpha=0.0, gamma=4.0, with_grad=False):
    logits, target = _f32(logits).reshape(-1), _f32(target).reshape(-1)
    grad = np.zeros_like(logits) if with_grad else None
    npos = C.c_double(0)
    val = lib().gga_oracle_focal_loss(logits, target, logits.size, alpha, gamma,
                                      grad.ctypes.data if with_grad else None, C.byref(npos))
    return (np.float32(val), grad, npos.value) if with_grad else (np.float32(val), npos.value)


def gather_pred(reg, height, dim, rot, ind):
    B, _, H, W = reg.shape
    K = ind.shape[1]
    pred = np.zeros((B, K, 8), np.float32)
    lib().gga_oracle_gather_pred(_f32(reg), _f32(height), _f32(dim), _f32(rot),
                                 np.ascontiguousarray(ind, np.int64), B, K, H, W, pred)
    return pred


def box_project(pred, ind, lidar2img, train_cfg):
    B, K, _ = pred.shape
    n = B * K
    fm_w = int(train_cfg['grid_size'][0]) // int(train_cfg['out_size_factor'])
    vs, pc = train_cfg['voxel_size'], train_cfg['point_cloud_range']
    rot = np.zeros(n, np.float32)
    ratio = np.zeros((n, 2), np.float32)
    iou = np.zeros((n, 4), np.float32)
    bev = np.zeros((n, 5), np.float32)
    lib().gga_oracle_box_project(_f32(pred).reshape(n, 8), np.ascontiguousarray(ind, np.int64).reshape(n),
                                 _f32(lidar2img).reshape(n, 16), n, fm_w, vs[0], vs[1],
                                 float(train_cfg['out_size_factor']), pc[0], pc[1], rot, ratio, iou, bev)
    return rot.reshape(B, K), ratio.reshape(B, K, 2), iou.reshape(B, K, 4), bev.reshape(B, K, 5)


def pal_object(points_f64, bev5):
    pts = np.ascontiguousarray(points_f64, np.float64)
    out = np.zeros(3, np.float32)
    lib().gga_oracle_pal_object(pts, pts.shape[0], pts.shape[1], _f32(bev5), out)
    return out


def l1_loss(pred, target, weight, avg_factor, loss_weight):
    p, t, w = (_f32(np.broadcast_to(a, pred.shape)).reshape(-1) for a in (pred, target, weight))
    return np.float32(lib().gga_oracle_l1_loss(p, t, w, p.size, np.float32(avg_factor), loss_weight))


def box_iou_rotated(b1, b2, mode='iou'):
    b1, b2 = _f32(b1).reshape(-1, 5), _f32(b2).reshape(-1, 5)
    out = np.zeros((len(b1), len(b2)), np.float32)
    for i in range(len(b1)):
        for j in range(len(b2)):
            out[i, j] = lib().gga_oracle_rotated_iou(b1[i], b2[j], int(mode == 'iof'))
    return out


def nms_rotated(boxes_xywhr, scores, thr):
    """mmcv.ops.nms_rotated: sort by descending score (stable), greedy suppression; returns keep
    indices into the input."""
    boxes, scores = _f32(boxes_xywhr).reshape(-1, 5), _f32(scores)
    order = np.argsort(-scores, kind='stable')
    keep = np.zeros(len(boxes), np.int64)
    n = lib().gga_oracle_nms_rotated_sorted(np.ascontiguousarray(boxes[order]), len(boxes), thr, keep)
    return order[keep[:n]]


def nms_bev(boxes_xyxyr, scores, thr, pre_max_size=None, post_max_size=None):
    """mmdet3d/core/post_processing/box3d_nms.py:231-268."""
    b, scores = _f32(boxes_xyxyr), _f32(scores)
    order = np.argsort(-scores, kind='stable')
    if pre_max_size is not None:
        order = order[:pre_max_size]
    b = b[order]
    xywhr = np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1],
                      b[:, 4]], -1)
    keep = order[nms_rotated(xywhr, scores[order], thr)]
    return keep[:post_max_size] if post_max_size is not None else keep


def points_in_boxes(points, boxes, all_boxes=False):
    pts, bx = _f32(points).reshape(-1, 3), _f32(boxes).reshape(-1, 7)
    out = np.zeros((len(pts), len(bx)) if all_boxes else (len(pts),), np.int32)
    lib().gga_oracle_points_in_boxes(pts, len(pts), bx, len(bx), int(all_boxes), out)
    return out


def image_box_overlap(boxes, query_boxes):
    """eval.py:86-114 (criterion -1); result in ``boxes.dtype`` like the reference."""
    b = np.ascontiguousarray(boxes, np.float64).reshape(-1, 4)
    q = np.ascontiguousarray(query_boxes, np.float64).reshape(-1, 4)
    out = np.zeros((len(b), len(q)), np.float64)
    if len(b) and len(q):
        lib().gga_oracle_image_box_overlap(b, len(b), q, len(q), int(np.asarray(boxes).dtype == np.float32), out)
    return out.astype(np.asarray(boxes).dtype)


def points_prepare(scene, sampled, centers_xy, min_distance, point_cloud_range):
    """One frame of the point pipeline tail (gga_processing.py:58-68,176; base_points.py:203-225):
    [sampled, scene minus points near a pasted centre], range-filtered, original order."""
    scene = _f32(scene); ndim = scene.shape[1]
    sampled = _f32(sampled).reshape(-1, ndim)
    ctr = np.ascontiguousarray(centers_xy, np.float64).reshape(-1, 2)
    out = np.zeros((len(scene) + len(sampled), ndim), np.float32)
    m = lib().gga_oracle_points_prepare(scene, len(scene), sampled, len(sampled), ctr, len(ctr), ndim, float(min_distance),
                                        _f32(point_cloud_range), out)
    return out[:m]


def region_grow(pc, mask_search, mask_origin, thresh, ratio=0.8):
    """tools/data_converter/utils_gga.py:6-38 (returns a float64 0/1 mask like the reference)."""
    pc = np.ascontiguousarray(pc, np.float64)
    ms = np.ascontiguousarray(np.asarray(mask_search) == 1, np.uint8)
    mo = np.ascontiguousarray(np.asarray(mask_origin) == 1, np.uint8)
    out = np.zeros(len(pc), np.uint8)
    lib().gga_oracle_region_grow(pc, len(pc), pc.shape[1], ms, mo, float(thresh), float(ratio if ratio is not None else 0.0),
                                 int(ratio is not None), out)
    return out.astype(np.float64)


def points_in_polyhedra(points, normal_vec, d):
    """box_np_ops.py:641-676 given surface_equ_3d's (normal_vec [P,S,3], d [P,S]) -> bool [N,P]."""
    pts = np.ascontiguousarray(points, np.float64)
    nv = np.ascontiguousarray(normal_vec, np.float64); dd = np.ascontiguousarray(d, np.float64)
    out = np.zeros((len(pts), nv.shape[0]), np.uint8)
    lib().gga_oracle_points_in_polyhedra(pts, len(pts), pts.shape[1], nv, dd, nv.shape[0], nv.shape[1], out)
    return out.astype(bool)


def plane_inliers(points, plane, thresh):
    pts = np.ascontiguousarray(points, np.float64)
    mask = np.zeros(len(pts), np.uint8)
    c = lib().gga_oracle_plane_inliers(pts, len(pts), pts.shape[1], np.ascontiguousarray(plane, np.float64), float(thresh), mask)
    return int(c), mask.astype(bool)


def frustum_normals(rect, Trv2c, P2, bbox):
    """Surface equations of the frustum of an image box in LiDAR coordinates, following
    utils_gga.py:87-98 and box_np_ops.py:13-33, 256-276, 526-552, 584-638 step by step."""
    CR, CT = P2[0:3, 0:3], P2[0:3, 3]
    Rinv, Cinv = np.linalg.qr(np.linalg.inv(CR))
    Cm, R, T = np.linalg.inv(Cinv), np.linalg.inv(Rinv), Cinv @ CT
    b = np.asarray(bbox).tolist()
    near_clip, far_clip = 0.001, 100
    fku, fkv, u0v0 = Cm[0, 0], -Cm[1, 1], Cm[0:2, 2]
    box = np.array([[b[0], b[1]], [b[0], b[3]], [b[2], b[3]], [b[2], b[1]]], dtype=Cm.dtype)
    near = (box - u0v0) / np.array([fku / near_clip, -fkv / near_clip], dtype=Cm.dtype)
    far = (box - u0v0) / np.array([fku / far_clip, -fkv / far_clip], dtype=Cm.dtype)
    fr = np.concatenate([np.concatenate([near, far], axis=0),
                         np.array([near_clip] * 4 + [far_clip] * 4, dtype=Cm.dtype)[:, None]], axis=1)
    fr -= T
    fr = (np.linalg.inv(R) @ fr.T).T
    fr = np.concatenate([fr, np.ones((8, 1))], axis=-1) @ np.linalg.inv((rect @ Trv2c).T)
    idx = np.array([0, 1, 2, 3, 7, 6, 5, 4, 0, 3, 7, 4, 1, 5, 6, 2, 0, 4, 5, 1, 3, 2, 6, 7]).reshape(6, 4)
    surf = fr[..., :3][idx][None]                                   # [1, 6, 4, 3]
    vec = surf[:, :, :2, :] - surf[:, :, 1:3, :]
    normal = np.cross(vec[:, :, 0, :], vec[:, :, 1, :])
    return normal, -np.einsum('aij, aij->ai', normal, surf[:, :, 0, :])


def points_in_frustm_indices(points, rect, Trv2c, P2, bbox):
    normal, d = frustum_normals(rect, Trv2c, P2, bbox)
    return points_in_polyhedra(np.ascontiguousarray(points[:, :3]), normal, d)


def calculate_ground(point_cloud, thresh_ransac=0.15, back_cut=False, back_cut_z=-5.0):
    """utils_gga.py:103-133 with the reference's np.random call sequence."""
    if back_cut:
        point_cloud = point_cloud[point_cloud[:, 2] > back_cut_z]
    cut = np.sort(point_cloud[:, 1])[int(point_cloud.shape[0] * 0.75)]
    cloud = point_cloud[point_cloud[:, 1] > cut]
    mask_all = np.ones(point_cloud.shape[0])
    final = None

    def collinear(p):
        a, b, c = np.linalg.norm(p[0] - p[1]), np.linalg.norm(p[1] - p[2]), np.linalg.norm(p[2] - p[0])
        h = (a + b + c) / 2
        return np.sqrt(h * (h - a) * (h - b) * (h - c)) < 1e-2

    for _ in range(5):
        best_len = 0
        for _it in range(min(cloud.shape[0], 100)):
            tri = cloud[np.random.choice(np.arange(cloud.shape[0]), size=(3), replace=False)]
            while collinear(tri):
                tri = cloud[np.random.choice(np.arange(cloud.shape[0]), size=(3), replace=False)]
            plane = np.linalg.solve(tri, np.ones(3))
            cnt, inl = plane_inliers(point_cloud, plane, thresh_ransac)
            if cnt > best_len and np.abs(np.dot(plane / np.linalg.norm(plane), np.array([0, 1, 0]))) > 0.9:
                mask_ground, best_len, final = inl, cnt, tri
        mask_all *= 1 - mask_ground
    return mask_all, final


def pseudo_label_match(dt_bboxes, gt_bboxes):
    """Per frame: argmax over ground truths of the detection/ground-truth image IoU
    (tools/utils_pseudo_labels_gga.py:44-59)."""
    return [np.argmax(image_box_overlap(d, g), axis=-1) if len(d) else np.zeros(0, np.int64)
            for d, g in zip(dt_bboxes, gt_bboxes)]


# ---------------------------------------------------------------------------
# CenterHead_GGA.get_targets / loss glue (centerpoint_head_gga.py:343-723)
# ---------------------------------------------------------------------------
SRL_PRIORS = ((1.35, 0.48), (3.60, 0.68), (2.40, 0.28))  # head:514-525 (Ped, Cyc, Car)


def draw_srl(n_frames, n_tasks=3, generator=None):
    """The reference draws ``clamp(N(mu_t, sigma_t), 1e-3)`` once per (frame, task)
    from the default CPU torch generator, frame-major / task-minor (head:514-525)."""
    import torch
    out = np.zeros((n_frames, n_tasks), np.float32)
    for b in range(n_frames):
        for t in range(n_tasks):
            mu, sd = SRL_PRIORS[t] if n_tasks == 3 else SRL_PRIORS[2]
            r = torch.normal(torch.tensor(mu), torch.tensor(sd), generator=generator)
            out[b, t] = float(torch.clamp(r, min=1e-3))
    return out


def get_targets(labels, boxes_img, lidar2img, pseudo, bdry, ibp, meta_l2i, train_cfg, srl,
                n_tasks=3):
    """Per-task targets for a batch. Inputs are per-frame lists of numpy arrays
    (``ibp[b]`` a list of [Ni,4] f64). ``srl`` [B, n_tasks] f32 are the SRL draws.
    Class ``t`` maps to task ``t`` (one class per task, head:426-434)."""
    B = len(labels)
    K = int(train_cfg['max_objs']) * int(train_cfg['dense_reg'])
    osf = train_cfg['out_size_factor']
    fw, fh = (int(g) // int(osf) for g in train_cfg['grid_size'][:2])
    vs = np.asarray(train_cfg['voxel_size'], np.float32)
    pc = np.asarray(train_cfg['point_cloud_range'], np.float32)
    T = n_tasks
    heat = np.zeros((T, B, 1, fh, fw), np.float32)
    anno = np.zeros((T, B, K, 5), np.float32)
    ind = np.zeros((T, B, K), np.int64)
    mask = np.zeros((T, B, K), np.uint8)
    l2i = np.zeros((T, B, K, 4, 4), np.float32)
    bmask = np.zeros((T, B, K, 4), np.uint8)
    ibps = [[[] for _ in range(B)] for _ in range(T)]
    for b in range(B):
        for t in range(T):
            l2i[t, b] = np.asarray(meta_l2i[b], np.float32)[None]      # head:508-509
            sel = np.flatnonzero(labels[b] == t)                       # head:426-434
            ibps[t][b] = [ibp[b][j] for j in sel]                      # head:469-470,481
            for k, j in enumerate(sel[:K]):
                pl = pseudo[b][j]                                      # f64
                # f64 / f32-tensor / python-int  (head:541-546)
                wg = pl[3] / np.float64(vs[0]) / osf
                lg = pl[4] / np.float64(vs[1]) / osf
                if not (wg > 0 and lg > 0):
                    continue
                r = gaussian_radius(lg, wg, train_cfg['gaussian_overlap'])
                r = max(int(train_cfg['min_radius']), int(r))
                cx = (pl[0] - np.float64(pc[0])) / np.float64(vs[0]) / osf     # head:557-563
                cy = (pl[1] - np.float64(pc[1])) / np.float64(vs[1]) / osf
                xi, yi = int(np.float32(cx)), int(np.float32(cy))      # f32 cast then trunc (:564-567)
                if not (0 <= xi < fw and 0 <= yi < fh):
                    continue
                draw_gaussian(heat[t, b, 0], xi, yi, r)
                ind[t, b, k] = yi * fw + xi
                mask[t, b, k] = 1
                l2i[t, b, k] = lidar2img[b][j]
                bmask[t, b, k] = ~np.asarray(bdry[b][j], bool)
                anno[t, b, k, :4] = boxes_img[b][j].astype(np.float32)
                anno[t, b, k, 4] = srl[b, t]
    return dict(heatmap=heat, anno_box=anno, ind=ind, mask=mask, lidar2img=l2i,
                bound_mask=bmask, ibp=ibps)


def head_loss(preds, tg, train_cfg, alpha=0.0, gamma=4.0, l1_weight=0.25):
    """CenterHead_GGA.loss (head:629-723) given per-task head maps ``preds[t]`` (dict of
    NCHW numpy arrays) and the targets of :func:`get_targets`. Returns the 18-entry dict
    plus the intermediate tensors per task."""
    losses, mids = {}, []
    cw = np.asarray(train_cfg['code_weights'], np.float32)
    for t, pd in enumerate(preds):
        B = pd['heatmap'].shape[0]
        lh, _ = focal_loss(pd['heatmap'], tg['heatmap'][t], alpha, gamma)
        ind, msk, anno = tg['ind'][t], tg['mask'][t], tg['anno_box'][t]
        pred = gather_pred(pd['reg'], pd['height'], pd['dim'], pd['rot'], ind)
        rot, ratio, iou, bev = box_project(pred, ind, tg['lidar2img'][t], train_cfg)
        K = ind.shape[1]
        p2c = np.zeros((3, B, K, 1), np.float32)                       # head:190-192
        for b in range(B):
            for k, pts in enumerate(tg['ibp'][t][b]):
                p2c[:, b, k, 0] = pal_object(pts, bev[b, k])
        num = np.float32(msk.astype(np.float32).sum())
        avg = np.float32(num + np.float32(1e-4))
        m5 = msk[..., None].astype(np.float32) * (~np.isnan(anno)).astype(np.float32)
        bw = m5 * cw                                                   # head:678-684
        zero = np.zeros_like(p2c[0])
        w0 = bw[..., 0:1]
        losses[f'task{t}.distancex'] = l1_loss(p2c[1], zero, w0, avg, l1_weight) * np.float32(0.1)
        losses[f'task{t}.distancey'] = l1_loss(p2c[2], zero, w0, avg, l1_weight) * np.float32(0.1)
        losses[f'task{t}.distancemin'] = l1_loss(p2c[0], zero, w0, avg, l1_weight) * np.float32(0.1)
        rw = np.minimum(ratio[..., 0:1], ratio[..., 1:2])
        rl = np.maximum(ratio[..., 0:1], ratio[..., 1:2])
        srl = rl - rw * anno[..., 4:5]                                 # head:703-711
        loss_srl = l1_loss(srl, np.zeros_like(srl), bw[..., 4:5], avg, l1_weight)
        wb = bw[..., :4] * tg['bound_mask'][t].astype(np.float32)      # head:714-717
        loss_bpl = l1_loss(iou, anno[..., :4], wb, avg, l1_weight)
        losses[f'task{t}.loss_heatmap'] = lh * np.float32(5.0)
        losses[f'task{t}.loss_bbox'] = loss_bpl * np.float32(0.3)
        losses[f'task{t}.loss_ratio'] = loss_srl * np.float32(0.1)
        mids.append(dict(pred=pred, rot=rot, pred_ratio=ratio, pred_iou=iou, pred_box_bev=bev,
                         p2c_min=p2c[0], p2c_x=p2c[1], p2c_y=p2c[2]))
    return losses, mids


def circle_nms(dets, thresh, post_max_size=83):
    """mmdet3d/core/post_processing/box3d_nms.py:181-225, loop for loop (numba there): dets [N,3]
    float32 (x, y, score); the distance is evaluated in the array's dtype, ``order`` is
    ``scores.argsort()[::-1]`` (stable here)."""
    dets = np.asarray(dets)
    x1, y1, scores = dets[:, 0], dets[:, 1], dets[:, 2]
    order = np.argsort(-scores, kind='stable').astype(np.int32)
    ndets = dets.shape[0]
    suppressed = np.zeros(ndets, dtype=np.int32)
    keep = []
    for _i in range(ndets):
        i = order[_i]
        if suppressed[i] == 1:
            continue
        keep.append(int(i))
        for _j in range(_i + 1, ndets):
            j = order[_j]
            if suppressed[j] == 1:
                continue
            dist = (x1[i] - x1[j]) ** 2 + (y1[i] - y1[j]) ** 2
            if dist <= thresh:
                suppressed[j] = 1
    return keep[:post_max_size] if post_max_size is not None else keep
