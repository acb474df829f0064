"""File-level driver of the offline GGA label generation - SURVEY.md §8(f) rank 3
(tools/data_converter/kitti_converter_gga.py:32-212 of the reference).

The reference fans ``_calculate_rga`` out over a 60-process pool (hours of CPU per dataset,
README.md:159), every worker dumping ``GGA_kitti_scene_<idx>.pkl``, and then merges the per-frame
files in ImageSets order. Here one process drives the device (``label_gen.calculate_rga``: a frame
is a handful of launches), with the same on-disk protocol: per-frame pickles (so an interrupted
run resumes), merged ``<prefix>_infos_<split>_GGA.pkl``. Input ``infos`` are the stock KITTI info
dicts (``get_kitti_image_info``: third-party-free file parsing of the raw dataset, not restated).
"""
import os
import pickle
from pathlib import Path

import numpy as np

from . import label_gen as LG
from .gt_database import points_in_rbbox


def limit_period(val, offset=0.5, period=np.pi):
    return val - np.floor(val / period + offset) * period


def box_camera_to_lidar(data, r_rect, velo2cam):
    """box_np_ops.box_camera_to_lidar (core/bbox/box_np_ops.py:45-67)."""
    xyz = data[:, 0:3]
    x_size, y_size, z_size = data[:, 3:4], data[:, 4:5], data[:, 5:6]
    r = data[:, 6:7]
    xyz_lidar = LG.camera_to_lidar(xyz, r_rect, velo2cam)
    r_new = limit_period(-r - np.pi / 2, period=np.pi * 2)
    return np.concatenate([xyz_lidar, x_size, z_size, y_size, r_new], axis=1)


def remove_outside_points(points, rect, Trv2c, P2, image_shape):
    """box_np_ops.remove_outside_points (:745-771): keep the points inside the camera frustum."""
    inside = LG.points_in_frustm_indices(points, rect, Trv2c, P2, [0, 0, image_shape[1], image_shape[0]])
    return points[inside.reshape([-1])]


def _velodyne_path(data_path, info, relative_path):
    p = info['point_cloud']['velodyne_path']
    return str(Path(data_path) / p) if relative_path else p


def calculate_num_points_in_gt(data_path, infos, relative_path, remove_outside=True, num_features=4):
    """``_calculate_num_points_in_gt`` (kitti_converter_gga.py:153-192): annos['num_points_in_gt']."""
    for info in infos:
        calib = info['calib']
        points_v = np.fromfile(_velodyne_path(data_path, info, relative_path), dtype=np.float32, count=-1).reshape([-1, num_features])
        rect, Trv2c, P2 = calib['R0_rect'], calib['Tr_velo_to_cam'], calib['P2']
        if remove_outside:
            points_v = remove_outside_points(points_v, rect, Trv2c, P2, info['image']['image_shape'])
        annos = info['annos']
        num_obj = len([n for n in annos['name'] if n != 'DontCare'])
        gt_boxes_camera = np.concatenate([annos['location'][:num_obj], annos['dimensions'][:num_obj],
                                          annos['rotation_y'][:num_obj][..., np.newaxis]], axis=1)
        gt_boxes_lidar = box_camera_to_lidar(gt_boxes_camera, rect, Trv2c)
        indices = points_in_rbbox(points_v[:, :3], gt_boxes_lidar)
        num_ignored = len(annos['dimensions']) - num_obj
        annos['num_points_in_gt'] = np.concatenate([indices.sum(0), -np.ones([num_ignored])]).astype(np.int32)
    return infos


def calculate_rga_file(data_path, info, relative_path, save_path, num_features=4, resume=False):
    """One frame of ``_calculate_rga`` including its file handling (:214-245, :517-518): read the
    velodyne file, run the label generation, dump ``GGA_kitti_scene_<idx>.pkl`` under ``save_path``."""
    filename = os.path.join(save_path, 'GGA_kitti_scene_{}.pkl'.format(info['image']['image_idx']))
    if resume and os.path.exists(filename):
        return filename
    points_v = np.fromfile(_velodyne_path(data_path, info, relative_path), dtype=np.float32, count=-1).reshape([-1, num_features])
    LG.calculate_rga(points_v, info['calib'], info['annos'], tuple(int(v) for v in info['image']['image_shape'][:2]))
    os.makedirs(save_path, exist_ok=True)
    with open(filename, 'wb') as f:
        pickle.dump(info, f)
    return filename


def create_gga_info_file(data_path, infos, image_ids, out_file, relative_path=True, save_path='./data/kitti_GGA_split_file',
                         resume=False, seed=None, logger=print, compute_num_points=True):
    """The GGA part of ``create_kitti_info_file`` for one split (:70-99): num_points_in_gt, the
    per-frame label generation, then the merge of the per-frame files in ``image_ids`` order into
    ``out_file``. ``seed``: re-seed ``np.random`` per frame with ``seed + image_idx`` (the RANSAC
    ground fit draws from it; the reference's pool workers inherit an arbitrary state)."""
    if compute_num_points:      # False: the infos already carry annos['num_points_in_gt'] (stock create_kitti_info_file)
        calculate_num_points_in_gt(data_path, infos, relative_path)
    for info in infos:
        if seed is not None:
            np.random.seed(seed + int(info['image']['image_idx']))
        calculate_rga_file(data_path, info, relative_path, save_path, resume=resume)
        logger('Finish Processing Sample {}'.format(info['image']['image_idx']))
    merged = []
    for idx in image_ids:
        with open(os.path.join(save_path, 'GGA_kitti_scene_{}.pkl'.format(int(idx))), 'rb') as f:
            merged.append(pickle.load(f))
    with open(out_file, 'wb') as f:
        pickle.dump(merged, f)
    logger(f'Kitti info file is saved to {out_file}')
    return merged
