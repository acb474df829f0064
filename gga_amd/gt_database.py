"""Ground-truth database of the GGA copy-paste augmentation - SURVEY.md §8(f) rank 3
(tools/data_converter/create_gt_database_gga.py:125-420 of the reference, LiDAR-only branch).

Per frame: every annotated object that passed the offline label generator
(``GGA_mask2d & GGA_mask_valid``) contributes the scene points inside the FRUSTUM of its 2D box
(not inside its 3D box: GGA has no 3D labels at this stage) as a ``.bin`` file, and one entry of
``<prefix>_dbinfos_train_GGA.pkl`` with the GGA side arrays the sampler re-attaches
(``DataBaseSampler_GGA``, gga_amd/pipelines.py).

The two point tests run on the device primitives of ``gga_amd.label_gen``
(``gga_points_in_convex_polyhedra``: float64, the reference's operation order):
frustum membership for the saved points, box membership for ``num_points_in_gt``
(``box_np_ops.points_in_rbbox``, core/bbox/box_np_ops.py:343-370). All boxes of a frame go through
ONE launch each. File handling and the info dicts are host code, as in the reference.
"""
import os
import pickle

import numpy as np

from . import label_gen as LG

DB_KEYS = ('GGA_gt_box', 'GGA_box_img', 'GGA_mask_depth', 'GGA_mask2d', 'GGA_mask_valid', 'GGA_mask_boundary',
           'GGA_bdry_mask', 'GGA_in_box_points', 'GGA_init_pseudo_label', 'GGA_num_points_in_box2d', 'GGA_lidar2img')


def points_in_rbbox(points, rbbox, z_axis=2, origin=(0.5, 0.5, 0)):
    """box_np_ops.points_in_rbbox: [N, M] bool, point n inside rotated box m."""
    if len(rbbox) == 0:
        return np.zeros((points.shape[0], 0), dtype=bool)
    corners = LG.center_to_corner_box3d(rbbox[:, :3], rbbox[:, 3:6], rbbox[:, 6], origin=origin, axis=z_axis)
    return LG.points_in_convex_polygon_3d(points[:, :3], LG.corner_to_surfaces_3d(corners))


def frame_entries(example, image_idx, info_prefix, group_counter, used_classes=None):
    """The database entries of one loaded frame (``example`` = the dict the dataset's loading
    pipeline returns: points, ann_info, rect / Trv2c / P2 / lidar2img).
    -> (list of (db_info, points [k, D] float32), next group counter)."""
    annos = example['ann_info']
    pts = example['points']
    points = pts.tensor.numpy() if hasattr(pts, 'tensor') else np.asarray(pts)
    gt_boxes_3d = annos['gt_bboxes_3d'].tensor.numpy()
    names = annos['gt_names']
    group_ids = annos['group_ids'] if 'group_ids' in annos else np.arange(gt_boxes_3d.shape[0], dtype=np.int64)
    difficulty = annos['difficulty'] if 'difficulty' in annos else np.zeros(gt_boxes_3d.shape[0], dtype=np.int32)
    m = annos['GGA_mask2d'] & annos['GGA_mask_valid']                     # filter invalid samples
    names, gt_boxes_3d, difficulty = names[m], gt_boxes_3d[m], difficulty[m]
    f = {k: annos[k][m] for k in ('bboxes', 'GGA_boxes_img', 'GGA_mask_depth', 'GGA_mask2d', 'GGA_mask_valid',
                                  'GGA_mask_boundary', 'GGA_bdry_masks', 'GGA_init_pseudo_label', 'GGA_num_points_in_box2d')}
    in_box = [item for item, keep in zip(annos['GGA_in_box_points'], m) if keep]
    num_obj = gt_boxes_3d.shape[0]
    assert f['GGA_boxes_img'].shape[0] == num_obj and f['bboxes'].shape[0] == num_obj, 'The boxes are mismatched'
    gt_point_indices = points_in_rbbox(points, gt_boxes_3d)
    point_indices = np.zeros((points.shape[0], num_obj), dtype=bool)
    live = [i for i, sign in enumerate(f['GGA_mask2d']) if sign]
    if live:       # the frusta of all live boxes in one launch
        surf = np.concatenate([LG.frustum_surfaces(example['rect'], example['Trv2c'], example['P2'], f['GGA_boxes_img'][i])
                               for i in live], 0)
        point_indices[:, live] = LG.points_in_convex_polygon_3d(points[:, :3], surf)
    out, group_dict = [], {}
    for i in range(num_obj):
        filename = f'{image_idx}_{names[i]}_{i}.bin'
        gga_points = points[point_indices[:, i]]                          # absolute coordinates (not box-relative)
        db_info = None
        if used_classes is None or names[i] in used_classes:
            db_info = {'name': names[i], 'path': os.path.join(f'{info_prefix}_gt_database_GGA', filename),
                       'image_idx': image_idx, 'gt_idx': i, 'box3d_lidar': gt_boxes_3d[i],
                       'num_points_in_gt': gt_point_indices[:, i].sum(), 'difficulty': difficulty[i],
                       'GGA_gt_box': f['bboxes'][i], 'GGA_box_img': f['GGA_boxes_img'][i],
                       'GGA_mask_depth': f['GGA_mask_depth'][i], 'GGA_mask2d': f['GGA_mask2d'][i],
                       'GGA_mask_valid': f['GGA_mask_valid'][i], 'GGA_mask_boundary': f['GGA_mask_boundary'][i],
                       'GGA_bdry_mask': f['GGA_bdry_masks'][i], 'GGA_in_box_points': in_box[i],
                       'GGA_init_pseudo_label': f['GGA_init_pseudo_label'][i],
                       'GGA_num_points_in_box2d': f['GGA_num_points_in_box2d'][i], 'GGA_lidar2img': example['lidar2img']}
            gid = group_ids[i]
            if gid not in group_dict:
                group_dict[gid] = group_counter
                group_counter += 1
            db_info['group_id'] = group_dict[gid]
            if 'score' in annos:
                db_info['score'] = annos['score'][i]
        out.append((filename, db_info, gga_points))
    return out, group_counter


def create_groundtruth_database(dataset, data_path, info_prefix, used_classes=None, database_save_path=None,
                                db_info_save_path=None, logger=print):
    """``create_groundtruth_database('KittiDataset_GGA', ...)`` for an already built dataset whose pipeline
    is ``[LoadPointsFromFile(load_dim=4, use_dim=4), LoadAnnotations3D(with_bbox_3d, with_label_3d)]``
    (create_gt_database_gga.py:148-171). Writes ``<prefix>_gt_database_GGA/*.bin`` and
    ``<prefix>_dbinfos_train_GGA.pkl``; returns the info dict."""
    database_save_path = database_save_path or os.path.join(data_path, f'{info_prefix}_gt_database_GGA')
    db_info_save_path = db_info_save_path or os.path.join(data_path, f'{info_prefix}_dbinfos_train_GGA.pkl')
    os.makedirs(database_save_path, exist_ok=True)
    all_db_infos, group_counter = {}, 0
    for j in range(len(dataset)):
        input_dict = dataset.get_data_info(j)
        dataset.pre_pipeline(input_dict)
        example = dataset.pipeline(input_dict)
        entries, group_counter = frame_entries(example, example['sample_idx'], info_prefix, group_counter, used_classes)
        for filename, db_info, gga_points in entries:
            with open(os.path.join(database_save_path, filename), 'w') as fh:
                gga_points.tofile(fh)
            if db_info is not None:
                all_db_infos.setdefault(db_info['name'], []).append(db_info)
    for k, v in all_db_infos.items():
        logger(f'load {len(v)} {k} database infos')
    with open(db_info_save_path, 'wb') as fh:
        pickle.dump(all_db_infos, fh)
    return all_db_infos
