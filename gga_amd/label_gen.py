"""Offline GGA label generation — SURVEY.md §8(f) rank 3.

Mirror of ``tools/data_converter/utils_gga.py`` of the reference (the numerical core of
``_calculate_rga``, ``tools/data_converter/kitti_converter_gga.py:214-517``, "hours of CPU per
dataset"): same function names, arguments and return values, the per-point work on the MI355X
through the C ABI (``gga_region_grow``, ``gga_points_in_convex_polyhedra``, ``gga_plane_inliers``),
float64 in the reference's operation order, so the masks are bit-identical to numpy's.

* ``region_grow(pc, mask_search, mask_origin, thresh, ratio)`` — utils_gga.py:6-38;
  ``region_grow_multi`` runs the seven thresholds the reference tries per object in one launch.
* ``points_in_frustm_indices(points, rect, Trv2c, P2, bbox_shape)`` — utils_gga.py:87-100. The
  8-corner frustum and its surface equations are a handful of 4x4 / 3x3 numpy operations taken in
  the reference's order (box_np_ops.py:13-33, 256-276, 526-552, 584-638); the point test runs on
  the device.
* ``calculate_ground(point_cloud, thresh_ransac, back_cut, back_cut_z)`` — utils_gga.py:103-133:
  the candidate triples are drawn on the host with ``np.random.choice`` in the reference's order
  (so equal seeds give equal planes), all candidates of a round are scored in one launch.
* ``fit_pseudo_box(cluster, ground_plane_height)`` — the initial pseudo 3D box of
  kitti_converter_gga.py:426-487 (minimum-area bounding rectangle over 36 headings).
* ``calculate_rga(points_v, calib, annos, image_shape)`` — the whole per-frame generator
  ``_calculate_rga`` as a function of the loaded frame (no file handling); ``box2d_labels``,
  ``view_points``, ``post_process_coords``, ``center_to_corner_box3d`` are its host-side pieces.
  Note that the reference's ``rotation_3d_in_axis`` computes numpy inputs in float32 (its
  ``array_converter``); ``_rotate_bev`` / ``_rotate3d`` reproduce that.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import functional as F
from ._lib import check

DEVICE = 'cuda:0'


def _dev(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype)).to(DEVICE)
    F._need_cuda(t)
    return t


# ----------------------------------------------------------------------------- region growing
def region_grow_multi(pc, mask_search, mask_origin, thresholds, ratio=0.8):
    """``region_grow`` for several distance thresholds at once -> float64 masks [T, N]."""
    pc = np.ascontiguousarray(pc, np.float64)
    n, dim = pc.shape
    th = np.ascontiguousarray(thresholds, np.float64).reshape(-1)
    if n == 0:
        return np.zeros((len(th), 0))
    d_pc = _dev(pc, np.float64)
    d_ms = _dev(np.asarray(mask_search) == 1, np.uint8)
    d_mo = _dev(np.asarray(mask_origin) == 1, np.uint8)
    d_th = _dev(th, np.float64)
    out = torch.empty((len(th), n), dtype=torch.uint8, device=DEVICE)
    L = _lib.lib()
    ws = F._workspace('region_grow', L.gga_region_grow_workspace_bytes(n, len(th)), torch.device(DEVICE))
    with torch.cuda.device(DEVICE):
        check(L.gga_region_grow(F._p(d_pc), n, dim, F._p(d_ms), F._p(d_mo), F._p(d_th), len(th),
                                float(ratio if ratio is not None else 0.0), int(ratio is not None), F._p(out), F._p(ws),
                                ws.numel(), F._stream()), 'gga_region_grow')
    return out.cpu().numpy().astype(np.float64)


def region_grow(pc, mask_search, mask_origin, thresh, ratio=0.8):
    return region_grow_multi(pc, mask_search, mask_origin, [thresh], ratio)[0]


# ----------------------------------------------------------------------------- frustum test
def projection_matrix_to_CRT_kitti(proj):
    """box_np_ops.py:526-552: P = C @ [R|T], C upper triangular (QR of the inverse)."""
    CR, CT = proj[0:3, 0:3], proj[0:3, 3]
    Rinv, Cinv = np.linalg.qr(np.linalg.inv(CR))
    C_ = np.linalg.inv(Cinv)
    return C_, np.linalg.inv(Rinv), Cinv @ CT


def get_frustum(bbox_image, C_, near_clip=0.001, far_clip=100):
    """box_np_ops.py:584-614: the 8 corners (4 near, 4 far) of the viewing frustum of an image box."""
    fku, fkv, u0v0 = C_[0, 0], -C_[1, 1], C_[0:2, 2]
    b = bbox_image
    corners = np.array([[b[0], b[1]], [b[0], b[3]], [b[2], b[3]], [b[2], b[1]]], dtype=C_.dtype)
    near = (corners - u0v0) / np.array([fku / near_clip, -fkv / near_clip], dtype=C_.dtype)
    far = (corners - u0v0) / np.array([fku / far_clip, -fkv / far_clip], dtype=C_.dtype)
    z = np.array([near_clip] * 4 + [far_clip] * 4, dtype=C_.dtype)[:, np.newaxis]
    return np.concatenate([np.concatenate([near, far], axis=0), z], axis=1)


def camera_to_lidar(points, r_rect, velo2cam):
    """box_np_ops.py:13-33."""
    if points.shape[-1] == 3:
        points = np.concatenate([points, np.ones(list(points.shape[:-1]) + [1])], axis=-1)
    return (points @ np.linalg.inv((r_rect @ velo2cam).T))[..., :3]


_SURFACE_CORNERS = np.array([0, 1, 2, 3, 7, 6, 5, 4, 0, 3, 7, 4, 1, 5, 6, 2, 0, 4, 5, 1, 3, 2, 6, 7]).reshape(6, 4)


def corner_to_surfaces_3d(corners):
    """box_np_ops.py:256-276: [N,8,3] -> [N,6,4,3], normals pointing inwards."""
    return corners[:, _SURFACE_CORNERS]


def surface_equ_3d(polygon_surfaces):
    """box_np_ops.py:617-638: (normal_vec, d) of a x + b y + c z + d = 0 per surface."""
    vec = polygon_surfaces[:, :, :2, :] - polygon_surfaces[:, :, 1:3, :]
    normal = np.cross(vec[:, :, 0, :], vec[:, :, 1, :])
    return normal, -np.einsum('aij, aij->ai', normal, polygon_surfaces[:, :, 0, :])


def points_in_convex_polygon_3d(points, polygon_surfaces):
    """box_np_ops.py:679-705 (all surfaces used) -> bool [N, P]; the point test is the device kernel."""
    normal, d = surface_equ_3d(polygon_surfaces[:, :, :3, :])
    pts = np.ascontiguousarray(points, np.float64)
    n, stride = pts.shape
    if n == 0:
        return np.zeros((0, normal.shape[0]), bool)
    d_pts, d_n, d_d = _dev(pts, np.float64), _dev(normal, np.float64), _dev(d, np.float64)
    out = torch.empty((n, normal.shape[0]), dtype=torch.uint8, device=DEVICE)
    with torch.cuda.device(DEVICE):
        check(_lib.lib().gga_points_in_convex_polyhedra(F._p(d_pts), n, stride, F._p(d_n), F._p(d_d), normal.shape[0],
                                                        normal.shape[1], F._p(out), F._stream()),
              'gga_points_in_convex_polyhedra')
    return out.cpu().numpy().astype(bool)


def frustum_surfaces(rect, Trv2c, P2, bbox_shape):
    C_, R, T = projection_matrix_to_CRT_kitti(P2)
    frustum = get_frustum(np.asarray(bbox_shape).tolist(), C_)
    frustum -= T
    frustum = np.linalg.inv(R) @ frustum.T
    frustum = camera_to_lidar(frustum.T, rect, Trv2c)
    return corner_to_surfaces_3d(frustum[np.newaxis, ...])


def points_in_frustm_indices(points, rect, Trv2c, P2, bbox_shape):
    """utils_gga.py:87-100 -> bool [N, 1]: LiDAR points inside the frustum of a 2D box."""
    return points_in_convex_polygon_3d(points[:, :3], frustum_surfaces(rect, Trv2c, P2, bbox_shape))


# ----------------------------------------------------------------------------- ground plane
def check_parallel(points):
    """utils_gga.py:41-52: (nearly) collinear triple, by Heron's formula."""
    a = np.linalg.norm(points[0] - points[1])
    b = np.linalg.norm(points[1] - points[2])
    c = np.linalg.norm(points[2] - points[0])
    p = (a + b + c) / 2
    return bool(np.sqrt(p * (p - a) * (p - b) * (p - c)) < 1e-2)


def fitPlane(points):
    """utils_gga.py:54-58: plane a.p = 1 through the points."""
    if points.shape[0] == points.shape[1]:
        return np.linalg.solve(points, np.ones(points.shape[0]))
    return np.linalg.lstsq(points, np.ones(points.shape[0]), rcond=-1)[0]


def calculate_ground(point_cloud, thresh_ransac=0.15, back_cut=False, back_cut_z=-5.0):
    """utils_gga.py:103-133 -> (mask_all: 1 = not ground, the triple that defined the last best plane)."""
    if back_cut:
        point_cloud = point_cloud[point_cloud[:, 2] > back_cut_z]
    temp = np.sort(point_cloud[:, 1])[int(point_cloud.shape[0] * 0.75)]
    cloud = point_cloud[point_cloud[:, 1] > temp]            # lowest quarter in camera coordinates (y down)
    pts = np.ascontiguousarray(point_cloud, np.float64)
    n = pts.shape[0]
    d_pts = _dev(pts, np.float64)
    mask_all = np.ones(n)
    final_sample_points = None
    L = _lib.lib()
    up = np.array([0, 1, 0])
    for _ in range(5):
        triples, planes = [], []
        for _it in range(min(cloud.shape[0], 100)):
            sampled = cloud[np.random.choice(np.arange(cloud.shape[0]), size=(3), replace=False)]
            while check_parallel(sampled):
                sampled = cloud[np.random.choice(np.arange(cloud.shape[0]), size=(3), replace=False)]
            triples.append(sampled)
            planes.append(fitPlane(sampled))
        if not planes:
            continue
        planes = np.ascontiguousarray(np.stack(planes), np.float64)
        counts = torch.empty(len(planes), dtype=torch.int32, device=DEVICE)
        masks = torch.empty((len(planes), n), dtype=torch.uint8, device=DEVICE)
        d_pl = _dev(planes, np.float64)
        with torch.cuda.device(DEVICE):
            check(L.gga_plane_inliers(F._p(d_pts), n, pts.shape[1], F._p(d_pl), len(planes), float(thresh_ransac), F._p(counts),
                                      F._p(masks), F._stream()), 'gga_plane_inliers')
        counts = counts.cpu().numpy()
        best_len, best = 0, -1
        for c, plane in enumerate(planes):                   # the reference's acceptance rule, in its order
            if counts[c] > best_len and np.abs(np.dot(plane / np.linalg.norm(plane), up)) > 0.9:
                best_len, best = counts[c], c
        if best >= 0:
            mask_ground = masks[best].cpu().numpy().astype(bool)
            final_sample_points = triples[best]
        mask_all *= 1 - mask_ground                          # NameError if no plane was ever accepted, like the reference
    return mask_all, final_sample_points


# ----------------------------------------------------------------------------- initial pseudo box
def _rotate_bev(xy, angle, clockwise):
    """structures/utils.py:28-117 for [N,2] numpy points and one angle. The reference's
    ``array_converter`` (core/utils/array_converter.py:283-299) turns numpy inputs into FLOAT32
    tensors, computes in float32 and casts the result back to the input dtype - reproduced here."""
    pts = torch.tensor(np.ascontiguousarray(xy), dtype=torch.float32)[None]
    ang = torch.full(pts.shape[:1], angle)
    s, c = torch.sin(ang), torch.cos(ang)
    rot_t = torch.stack([torch.stack([c, s]), torch.stack([-s, c])])
    if clockwise:
        rot_t = rot_t.transpose(0, 1)
    return torch.einsum('aij,jka->aik', pts, rot_t)[0].numpy().astype(np.asarray(xy).dtype)


def fit_pseudo_box(cur_clt, ground_plane_height):
    """kitti_converter_gga.py:426-487: the minimum-area BEV rectangle over the headings
    0, 2.5, ..., 87.5 degrees, longer side first, z from the cluster top down to the ground plane.
    -> (pseudo_bbox_3d [1,7], centre [1,2], heading)."""
    rot_list = np.arange(0, (np.pi / 2.0 - 1e-6), np.pi / 72.0).tolist()
    area, centre, edge = [], [], []
    for r in rot_list:
        q = _rotate_bev(cur_clt[..., :2], r, clockwise=True)
        x0, x1, y0, y1 = np.min(q[..., 0]), np.max(q[..., 0]), np.min(q[..., 1]), np.max(q[..., 1])
        area.append((x1 - x0) * (y1 - y0))
        centre.append(np.array([(x0 + x1) / 2.0, (y0 + y1) / 2.0]))
        edge.append(np.array([x1 - x0, y1 - y0]))
    sel = np.argsort(np.array(area))[0]
    sel_rot = rot_list[sel]
    sel_edge = np.stack(edge)[sel, None]
    sel_center_ori = _rotate_bev(np.stack(centre)[sel, None], sel_rot, clockwise=False)
    if sel_edge[:, 0] < sel_edge[:, 1]:
        sel_edge = sel_edge[:, ::-1]
        sel_rot = sel_rot + np.pi / 2.0
    top = np.max(cur_clt[:, 2])
    zc = np.array((top + ground_plane_height) / 2.0)[np.newaxis]
    dz = np.array(top - ground_plane_height)[np.newaxis]
    box = np.concatenate([sel_center_ori.squeeze(), zc, sel_edge.squeeze(), dz, np.array(sel_rot)[np.newaxis]])[np.newaxis]
    return box, sel_center_ori, sel_rot


# ----------------------------------------------------------------------------- whole-frame label generation
def view_points(points, view, normalize):
    """nuscenes-devkit ``geometry_utils.view_points`` (third-party, published): [3,n] points through
    a <=4x4 view matrix, optionally divided by depth."""
    viewpad = np.eye(4)
    viewpad[:view.shape[0], :view.shape[1]] = view
    n = points.shape[1]
    p = np.dot(viewpad, np.concatenate((points, np.ones((1, n)))))[:3, :]
    if normalize:
        p = p / p[2:3, :].repeat(3, 0).reshape(3, n)
    return p


def _convex_hull(pts):
    """Andrew's monotone chain, counter-clockwise, collinear points dropped."""
    pts = sorted(set(map(tuple, pts)))
    if len(pts) <= 2:
        return pts
    cross = lambda o, a, b: (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lo, up = [], []
    for p in pts:
        while len(lo) >= 2 and cross(lo[-2], lo[-1], p) <= 0:
            lo.pop()
        lo.append(p)
    for p in reversed(pts):
        while len(up) >= 2 and cross(up[-2], up[-1], p) <= 0:
            up.pop()
        up.append(p)
    return lo[:-1] + up[:-1]


def post_process_coords(corner_coords, imsize=(1600, 900)):
    """tools/data_converter/nuscenes_converter.py:534-564 without shapely (absent here): bounding
    box of (convex hull of the projected corners) ∩ (image canvas), None when they do not meet.
    Hull by monotone chain, intersection by Sutherland-Hodgman clipping - parity with GEOS unpinned."""
    hull = _convex_hull(corner_coords)
    if len(hull) < 3:
        return None
    poly = hull
    for axis, bound, keep_less in ((0, 0.0, False), (0, float(imsize[0]), True), (1, 0.0, False), (1, float(imsize[1]), True)):
        out = []
        for i, cur in enumerate(poly):
            prev = poly[i - 1]
            cin = cur[axis] <= bound if keep_less else cur[axis] >= bound
            pin = prev[axis] <= bound if keep_less else prev[axis] >= bound
            if cin != pin:
                t = (bound - prev[axis]) / (cur[axis] - prev[axis])
                q = [prev[0] + t * (cur[0] - prev[0]), prev[1] + t * (cur[1] - prev[1])]
                q[axis] = bound
                out.append(tuple(q))
            if cin:
                out.append(cur)
        poly = out
        if not poly:
            return None
    xs, ys = [p[0] for p in poly], [p[1] for p in poly]
    return min(xs), min(ys), max(xs), max(ys)


def _rotate3d(points, angles, axis, clockwise=False):
    """structures/utils.py:28-117 for [N,M,3] points and [N] angles given as numpy arrays: float32
    arithmetic, result cast back to the points' dtype (see ``_rotate_bev``)."""
    pts = torch.tensor(np.ascontiguousarray(points), dtype=torch.float32)
    ang = torch.tensor(np.ascontiguousarray(angles), dtype=torch.float32)
    s, c = torch.sin(ang), torch.cos(ang)
    one, zero = torch.ones_like(c), torch.zeros_like(c)
    if axis == 1:
        rot = torch.stack([torch.stack([c, zero, -s]), torch.stack([zero, one, zero]), torch.stack([s, zero, c])])
    elif axis == 2:
        rot = torch.stack([torch.stack([c, s, zero]), torch.stack([-s, c, zero]), torch.stack([zero, zero, one])])
    else:
        rot = torch.stack([torch.stack([one, zero, zero]), torch.stack([zero, c, s]), torch.stack([zero, -s, c])])
    if clockwise:
        rot = rot.transpose(0, 1)
    return torch.einsum('aij,jka->aik', pts, rot).numpy().astype(np.asarray(points).dtype)


def center_to_corner_box3d(centers, dims, angles=None, origin=(0.5, 1.0, 0.5), axis=1):
    """box_np_ops.py:62-93,170-200: [N,8,3] corners of boxes given centres, sizes and yaw."""
    norm = np.stack(np.unravel_index(np.arange(8), [2] * 3), axis=1).astype(dims.dtype)[[0, 1, 3, 2, 4, 5, 7, 6]]
    corners = dims.reshape([-1, 1, 3]) * (norm - np.array(origin, dtype=dims.dtype)).reshape([1, 8, 3])
    if angles is not None:
        corners = _rotate3d(corners, angles, axis=axis)
    corners += centers.reshape([-1, 1, 3])
    return corners


def project_pts_on_img(points, img_shape, lidar2img):
    """utils_gga.py:61-84 (only the image shape is used of the image) -> (pixels, points, in-image mask)."""
    pts_2d = np.concatenate([points[:, :3], np.ones((points.shape[0], 1))], axis=-1) @ lidar2img.T
    pts_2d[:, 2] = np.clip(pts_2d[:, 2], a_min=1e-5, a_max=99999)
    pts_2d[:, 0] /= pts_2d[:, 2]
    pts_2d[:, 1] /= pts_2d[:, 2]
    pix = np.round(pts_2d[:, :2]).astype(np.int64)
    fov = (pix[:, 0] < img_shape[1]) & (pix[:, 0] >= 0) & (pix[:, 1] < img_shape[0]) & (pix[:, 1] >= 0)
    return pix[fov, :3], points[fov], fov


def box2d_labels(annos, P2, image_shape):
    """2D-box part of _calculate_rga (kitti_converter_gga.py:252-326): per object the image box of
    its projected 3D box, and whether it is in the image / fully in front / away from the border."""
    n_obj = len([n for n in annos['name'] if n != 'DontCare'])
    boxes = np.concatenate([annos['location'][:n_obj], annos['dimensions'][:n_obj], annos['rotation_y'][:n_obj, np.newaxis]], axis=1)
    img_size = (image_shape[1] - 1, image_shape[0] - 1)
    img_boundary = np.array([0, 0, img_size[0], img_size[1]])
    mask2d, box2d, depth_mask, bdry_masks, mask_boundary = [], [], [], [], []
    for box3d in boxes:
        box3d = box3d[np.newaxis, :]
        corners = center_to_corner_box3d(box3d[:, :3], box3d[:, 3:6], box3d[:, 6], [0.5, 1.0, 0.5], axis=1)[0].T
        in_front = np.argwhere(corners[2, :] > 0).flatten()
        corners = corners[:, in_front]
        coords = view_points(corners, P2, True).T[:, :2].tolist()
        final = post_process_coords(coords, img_size)
        if final is None:
            mask2d.append(False); depth_mask.append(False); mask_boundary.append(False)
            box2d.append(np.array(-np.ones([1, 4])))
            bdry_masks.append(np.ones(4).astype(np.bool_))
        else:
            mask2d.append(True)
            depth_mask.append(in_front.shape[0] == 8)
            final = np.array(final)[np.newaxis, :]
            box2d.append(final)
            bm = final[0] == img_boundary
            bdry_masks.append(bm)
            mask_boundary.append(np.all(~bm))
    return (np.concatenate(box2d), np.array(depth_mask), np.array(mask2d), np.array(mask_boundary), np.stack(bdry_masks))


def calculate_rga(points_v, calib, annos, image_shape):
    """The per-frame GGA label generation of ``_calculate_rga`` (kitti_converter_gga.py:214-517) as a
    function of the loaded frame: ``points_v`` [N, >=3] f32 velodyne points, ``calib`` with 4x4
    ``R0_rect`` / ``Tr_velo_to_cam`` / ``P2``, KITTI ``annos`` (modified in place and returned),
    ``image_shape`` (H, W). Adds GGA_boxes_img / GGA_mask_* / GGA_bdry_masks / GGA_in_box_points /
    GGA_init_pseudo_label / GGA_num_points_in_box2d exactly as the reference lays them out
    (DontCare entries padded at the end). Ground removal draws from ``np.random`` like the reference."""
    rect, Trv2c, P2 = calib['R0_rect'], calib['Tr_velo_to_cam'], calib['P2']
    points_lidar = points_v[..., :3]
    points_lidar = np.concatenate([points_lidar, np.ones(list(points_lidar.shape[0:-1]) + [1])], axis=-1)
    points_cam = points_lidar @ (rect @ Trv2c).T
    mask_ground_all, _ = calculate_ground(points_cam[..., :3], 0.2)
    ground_plane_height = points_lidar[(1 - mask_ground_all).astype(np.bool_)][:, 2].mean()

    num_obj = len([n for n in annos['name'] if n != 'DontCare'])
    name = annos['name'][:num_obj]
    num_points_in_gt = annos['num_points_in_gt']
    gt_boxes_img, depth_mask, mask2d, mask_boundary, bdry_masks = box2d_labels(annos, P2, image_shape)
    annos['GGA_boxes_img'] = gt_boxes_img
    annos['GGA_mask_depth'], annos['GGA_mask2d'] = depth_mask, mask2d
    annos['GGA_mask_boundary'], annos['GGA_bdry_masks'] = mask_boundary, bdry_masks

    lidar2img = P2 @ rect @ Trv2c
    object_filter_all = project_pts_on_img(points_lidar, image_shape, lidar2img)[2]

    # objects front to back (median depth of the points inside their frustum)
    boxes_img = gt_boxes_img.copy()
    frusta = [points_in_frustm_indices(points_lidar, rect, Trv2c, P2, b).squeeze() for b in boxes_img]
    isvalid, medis = [], []
    for index, bpi in enumerate(frusta):
        if (bpi.sum() == 0) or (num_points_in_gt[index] == 0):
            medis.append(1000); isvalid.append(False)
        else:
            medis.append(np.median(points_cam[bpi][:, 2])); isvalid.append(True)
    obj_ord = np.argsort(np.array(medis))

    points_cluster = []
    mask_object = np.ones((points_lidar.shape[0]))
    filter_z = points_cam[:, 2] > 0
    thresholds = [(j + 1) * 0.1 for j in range(7)]
    for element in np.nditer(obj_ord):
        element = int(element)
        if not isvalid[element]:
            points_cluster.append(np.array([]))
            continue
        ratio = 0.96 if name[element] == 'Car' else 0.85
        mask_search = mask_ground_all * object_filter_all * mask_object * filter_z
        mask_origin = mask_ground_all * frusta[element] * mask_object * filter_z
        segs = region_grow_multi(points_cam, mask_search, mask_origin, thresholds, ratio)   # the 7 calls share their masks
        result = np.zeros((7, 2))
        count = 0
        kept = []
        for j in range(7):
            seg = segs[j]
            if seg.sum() == 0:
                continue
            if j >= 1:
                old = kept[-1]                           # IndexError like the reference if the first hit is not j = 0
                if old.sum() != (seg * old).sum():
                    count += 1
            result[count, 0] = j
            result[count, 1] = seg.sum()
            kept.append(seg)
        best_j = result[np.argmax(result[:, 1]), 0]
        try:
            best = kept[int(best_j)]                     # the reference indexes its list of non-empty results by threshold index
            mask_object *= (1 - best)
            pc = points_lidar[best == 1].copy()
            if annos['GGA_mask_boundary'][element] == True:      # noqa: E712
                points_cluster.append(pc)
        except IndexError:
            points_cluster.append(np.array([]))
            continue
        if annos['GGA_mask_boundary'][element] == False:         # noqa: E712  (box cut by the image border: grow past the frustum)
            grown = region_grow(points_cam, mask_ground_all, best, (best_j + 1) * 0.1, ratio=None)
            pc_truncate = points_lidar[grown == 1].copy()
            points_cluster.append(pc if pc_truncate.shape[0] > 6000 else pc_truncate)

    point_cluster_ord = [points_cluster[int(pos)] for pos in np.nditer(np.argsort(obj_ord))] if len(obj_ord) else []

    pseudo, n_in, valid = [], [], []
    for clt in point_cluster_ord:
        if clt.shape[0] == 0:
            n_in.append(0); valid.append(False); pseudo.append(np.zeros([1, 7]))
            continue
        box, _, _ = fit_pseudo_box(clt, ground_plane_height)
        n_in.append(clt.shape[0]); valid.append(True); pseudo.append(box)

    annos['GGA_mask_valid'] = np.stack(valid)
    annos['GGA_in_box_points'] = point_cluster_ord
    annos['GGA_init_pseudo_label'] = np.concatenate(pseudo)
    annos['GGA_num_points_in_box2d'] = np.array(n_in)
    n_ign = len(annos['dimensions']) - num_obj
    annos['GGA_boxes_img'] = np.concatenate((annos['GGA_boxes_img'], -np.zeros([n_ign, 4])), axis=0)
    for key in ('GGA_mask2d', 'GGA_mask_depth', 'GGA_mask_boundary', 'GGA_mask_valid'):
        annos[key] = np.concatenate((annos[key], np.zeros([n_ign]).astype(bool)))
    annos['GGA_num_points_in_box2d'] = np.concatenate((annos['GGA_num_points_in_box2d'], np.zeros([n_ign])))
    annos['GGA_init_pseudo_label'] = np.concatenate((annos['GGA_init_pseudo_label'], np.zeros([n_ign, 7])), axis=0)
    annos['GGA_bdry_masks'] = np.concatenate((annos['GGA_bdry_masks'], np.zeros([n_ign, 4]).astype(bool)))
    annos['GGA_in_box_points'].extend([np.array([]) for _ in range(n_ign)])
    return annos
