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
    """structures/utils.py:28-117 for [N,2] points and one angle; note that the reference builds the
    angle tensor with ``torch.full`` (float32), so sin / cos are float32 values."""
    pts = torch.from_numpy(np.ascontiguousarray(xy))[None]
    ang = torch.full(pts.shape[:1], angle)
    s, c = torch.sin(ang), torch.cos(ang)
    rot_t = torch.stack([torch.stack([c, s]), torch.stack([-s, c])])
    if clockwise:
        rot_t = rot_t.transpose(0, 1)
    return torch.einsum('aij,jka->aik', pts, rot_t.to(pts.dtype))[0].numpy()


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
