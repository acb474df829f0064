"""``KittiDataset_GGA_train`` and the annotation loader of the GGA train pipeline - SURVEY.md §8(f)
rank 2, the data format on the input side of the hot path.

Mirrors (same names, arguments, dict keys and dtypes)

* ``KittiDataset_GGA_train.get_data_info / get_ann_info / remove_dontcare_GGA / drop_arrays_by_name``
  (mmdet3d/datasets/kitti_dataset_GGA_train.py:100-329) and the ``Custom3DDataset`` plumbing they
  rely on (``load_annotations``, ``pre_pipeline``, ``prepare_train_data``, ``__getitem__``:
  mmdet3d/datasets/custom_3d.py) for local ``.pkl`` info files;
* ``LoadAnnotations3D`` with ``with_gga=True`` (mmdet3d/datasets/pipelines/loading.py:560-691):
  ``_load_bboxes_3d``, ``_load_labels_3d``, ``_load_GGA_labels``;
* the camera -> LiDAR box conversion the loader goes through
  (``CameraInstance3DBoxes.convert_to`` = ``Box3DMode.convert``, core/bbox/structures/box_3d_mode.py:63-199).

Pinned by tests/test_datasets.py against a run of the reference's own classes on three frames
(tests/golden/gt_database.npz, tools_dev/make_golden.py::golden_gt_database).
"""
import copy
import os
import pickle

import numpy as np
import torch

from .box3d import LiDARInstance3DBoxes
from .pipelines import Compose
from .registry import DATASETS, PIPELINES


def limit_period(val, offset=0.5, period=np.pi):
    """core/bbox/structures/utils.py:11-25."""
    return val - torch.floor(val / period + offset) * period


def camera_boxes_to_lidar(boxes, rt_mat):
    """``Box3DMode.convert(box, CAM, LIDAR, rt_mat)`` for an [N,7] float32 array
    (x, y, z, x_size, y_size, z_size, yaw) in camera coordinates (box_3d_mode.py:113-168): centres
    through ``rt_mat``, sizes (l, h, w) -> (l, w, h), yaw -> limit_period(-yaw - pi/2, 2 pi). torch
    arithmetic like the reference (float32 boxes, the matrix cast to their dtype)."""
    arr = torch.from_numpy(np.asarray(boxes)).clone()
    x_size, y_size, z_size, yaw = arr[..., 3:4], arr[..., 4:5], arr[..., 5:6], arr[..., 6:7]
    xyz_size = torch.cat([x_size, z_size, y_size], dim=-1)
    yaw = -yaw - np.pi / 2
    yaw = limit_period(yaw, period=np.pi * 2)
    rt = arr.new_tensor(np.asarray(rt_mat))
    if rt.size(1) == 4:
        ext = torch.cat([arr[..., :3], arr.new_ones(arr.size(0), 1)], dim=-1)
        xyz = ext @ rt.t()
    else:
        xyz = arr[..., :3] @ rt.t()
    return torch.cat([xyz[..., :3], xyz_size, yaw, arr[..., 7:]], dim=-1)


@DATASETS.register_module()
class KittiDataset_GGA_train:
    CLASSES = ('car', 'pedestrian', 'cyclist')

    def __init__(self, data_root, ann_file, split, pts_prefix='velodyne', pipeline=None, classes=None, modality=None,
                 box_type_3d='LiDAR', filter_empty_gt=True, test_mode=False,
                 pcd_limit_range=[0, -40, -3, 70.4, 40, 0.0], **kwargs):
        self.data_root, self.ann_file = data_root, ann_file
        self.test_mode, self.modality, self.filter_empty_gt = test_mode, modality, filter_empty_gt
        if box_type_3d.lower() != 'lidar':
            raise NotImplementedError('the GGA LiDAR pipeline uses box_type_3d="LiDAR"')
        self.box_type_3d, self.box_mode_3d = LiDARInstance3DBoxes, 0       # Box3DMode.LIDAR
        self.CLASSES = tuple(classes) if classes is not None else self.CLASSES
        self.cat2id = {name: i for i, name in enumerate(self.CLASSES)}
        self.data_infos = self.load_annotations(self.ann_file)
        self.pipeline = Compose(pipeline) if pipeline is not None else None
        self.split = split
        self.root_split = os.path.join(self.data_root, split)
        assert self.modality is not None
        self.pcd_limit_range, self.pts_prefix = pcd_limit_range, pts_prefix
        if not test_mode:            # sampler groups (aspect-ratio flags of the image datasets): one group here
            self.flag = np.zeros(len(self), dtype=np.uint8)

    def load_annotations(self, ann_file):
        if isinstance(ann_file, (list, tuple)):         # already loaded infos
            return list(ann_file)
        with open(ann_file, 'rb') as f:
            return pickle.load(f)

    def __len__(self):
        return len(self.data_infos)

    def _get_pts_filename(self, idx):
        return os.path.join(self.root_split, self.pts_prefix, f'{idx:06d}.bin')

    def get_data_info(self, index):
        info = self.data_infos[index]
        sample_idx = info['image']['image_idx']
        img_filename = os.path.join(self.data_root, info['image']['image_path'])
        rect = info['calib']['R0_rect'].astype(np.float32)
        Trv2c = info['calib']['Tr_velo_to_cam'].astype(np.float32)
        P2 = info['calib']['P2'].astype(np.float32)
        lidar2img = P2 @ rect @ Trv2c
        input_dict = dict(sample_idx=sample_idx, pts_filename=self._get_pts_filename(sample_idx), img_prefix=None,
                          img_info=dict(filename=img_filename), lidar2img=lidar2img, rect=rect, Trv2c=Trv2c, P2=P2)
        if not self.test_mode:
            input_dict['ann_info'] = self.get_ann_info(index)
        return input_dict

    def get_ann_info(self, index):
        info = self.data_infos[index]
        rect = info['calib']['R0_rect'].astype(np.float32)
        Trv2c = info['calib']['Tr_velo_to_cam'].astype(np.float32)
        plane_lidar = None
        if 'plane' in info:          # ground plane (n, d) of the camera frame -> LiDAR frame: normal rotated, a point of it moved
            cam2lidar = np.linalg.inv(rect @ Trv2c)
            R, t = cam2lidar[:3, :3], cam2lidar[:3, 3]
            normal_cam = info['plane'][:3]
            on_plane_cam = -normal_cam * info['plane'][3]
            normal = (R @ normal_cam[:, None])[:, 0]
            on_plane = R @ on_plane_cam[:, None][:, 0] + t
            plane_lidar = np.zeros_like(normal, shape=(4, ))
            plane_lidar[:3], plane_lidar[3] = normal, -normal.T @ on_plane
        annos = self.remove_dontcare_GGA(info['annos'])      # other objects stay: collision tests when sampling
        difficulty = annos['difficulty']
        gt_names = annos['name']
        gt_bboxes_3d = np.concatenate([annos['location'], annos['dimensions'], annos['rotation_y'][..., np.newaxis]],
                                      axis=1).astype(np.float32)
        gt_bboxes_3d = LiDARInstance3DBoxes(camera_boxes_to_lidar(gt_bboxes_3d, np.linalg.inv(rect @ Trv2c)))
        gt_bboxes = annos['bbox']
        selected = self.drop_arrays_by_name(gt_names, ['DontCare'])
        gt_bboxes = gt_bboxes[selected].astype('float32')
        gt_names = gt_names[selected]
        gga = {k: annos[k][selected] for k in ('GGA_boxes_img', 'GGA_mask_depth', 'GGA_mask2d', 'GGA_mask_valid',
                                                'GGA_mask_boundary', 'GGA_bdry_masks', 'GGA_init_pseudo_label',
                                                'GGA_num_points_in_box2d')}
        in_box = [annos['GGA_in_box_points'][i] for i in selected.tolist()]
        gt_labels = np.array([self.CLASSES.index(cat) if cat in self.CLASSES else -1 for cat in gt_names]).astype(np.int64)
        return dict(gt_bboxes_3d=gt_bboxes_3d, gt_labels_3d=copy.deepcopy(gt_labels), bboxes=gt_bboxes, labels=gt_labels,
                    gt_names=gt_names, plane=plane_lidar, difficulty=difficulty, GGA_boxes_img=gga['GGA_boxes_img'],
                    GGA_mask_depth=gga['GGA_mask_depth'], GGA_mask2d=gga['GGA_mask2d'], GGA_mask_valid=gga['GGA_mask_valid'],
                    GGA_mask_boundary=gga['GGA_mask_boundary'], GGA_bdry_masks=gga['GGA_bdry_masks'],
                    GGA_in_box_points=in_box, GGA_init_pseudo_label=gga['GGA_init_pseudo_label'],
                    GGA_num_points_in_box2d=gga['GGA_num_points_in_box2d'])

    def drop_arrays_by_name(self, gt_names, used_classes):
        return np.array([i for i, x in enumerate(gt_names) if x not in used_classes], dtype=np.int64)

    def keep_arrays_by_name(self, gt_names, used_classes):
        return np.array([i for i, x in enumerate(gt_names) if x in used_classes], dtype=np.int64)

    def remove_dontcare_GGA(self, ann_info):
        keep = [i for i, x in enumerate(ann_info['name']) if x != 'DontCare']
        out = {}
        for key, val in ann_info.items():          # list-valued fields (the in-box point sets) filtered by hand
            out[key] = [val[i] for i in keep] if isinstance(val, list) else val[keep]
        return out

    # ---- sample plumbing (the hooks of Custom3DDataset: custom_3d.py:127-156, 204-245, 430-448)
    _FIELD_REGISTRIES = ('img_fields', 'bbox3d_fields', 'pts_mask_fields', 'pts_seg_fields', 'bbox_fields', 'mask_fields', 'seg_fields')

    def pre_pipeline(self, results):
        """What every transform expects to find in a fresh sample: the (empty) registries of field names they append to, and
        the box class / coordinate mode of this dataset."""
        for registry in self._FIELD_REGISTRIES:
            results[registry] = []
        results.update(box_type_3d=self.box_type_3d, box_mode_3d=self.box_mode_3d)

    def _through_pipeline(self, index):
        sample = self.get_data_info(index)
        if sample is not None:
            self.pre_pipeline(sample)
            sample = self.pipeline(sample)
        return sample

    def prepare_train_data(self, index):
        """The train sample of ``index``, or None when there is nothing to learn from it (no info, a transform gave up, or -
        with ``filter_empty_gt`` - no object of a known class is left after the filters)."""
        example = self._through_pipeline(index)
        if example is None:
            return None
        if self.filter_empty_gt and not bool((_unwrap(example['gt_labels_3d']) != -1).any()):
            return None
        return example

    def _rand_another(self, idx):
        return np.random.choice(len(self))

    def __getitem__(self, idx):
        if self.test_mode:
            return self._through_pipeline(idx)
        data = self.prepare_train_data(idx)
        while data is None:                      # an unusable frame is replaced by a random other one
            data = self.prepare_train_data(self._rand_another(idx))
        return data


def _unwrap(x):
    return getattr(x, '_data', getattr(x, 'data', x))


@PIPELINES.register_module()
class LoadAnnotations3D:
    """loading.py:560-691 for the keys of the GGA LiDAR pipeline: 3D boxes, 3D labels and - with
    ``with_gga=True`` - the GGA side arrays (``_load_GGA_labels``, loading.py:650-661)."""

    def __init__(self, with_bbox_3d=True, with_label_3d=True, with_attr_label=False, with_mask_3d=False, with_seg_3d=False,
                 with_bbox=False, with_label=False, with_mask=False, with_seg=False, with_bbox_depth=False, with_gga=False,
                 poly2mask=True, seg_3d_dtype=np.int64, file_client_args=dict(backend='disk')):
        for flag, name in ((with_attr_label, 'with_attr_label'), (with_mask_3d, 'with_mask_3d'), (with_seg_3d, 'with_seg_3d'),
                           (with_mask, 'with_mask'), (with_seg, 'with_seg'), (with_bbox_depth, 'with_bbox_depth')):
            if flag:
                raise NotImplementedError(f'LoadAnnotations3D({name}=True) is not on the GGA LiDAR path')
        self.with_bbox_3d, self.with_label_3d, self.with_bbox, self.with_label = with_bbox_3d, with_label_3d, with_bbox, with_label
        self.with_gga = with_gga

    def _load_bboxes_3d(self, results):
        results['gt_bboxes_3d'] = results['ann_info']['gt_bboxes_3d']
        results['bbox3d_fields'].append('gt_bboxes_3d')
        return results

    def _load_labels_3d(self, results):
        results['gt_labels_3d'] = results['ann_info']['gt_labels_3d']
        return results

    def _load_GGA_labels(self, results):
        a = results['ann_info']
        results['GGA_boxes_img'] = a['GGA_boxes_img']
        results['GGA_lidar2img'] = results['lidar2img'][np.newaxis, ...].repeat(len(results['GGA_boxes_img']), axis=0)
        results['GGA_init_pseudo_labels'] = a['GGA_init_pseudo_label']
        results['GGA_in_box_points'] = a['GGA_in_box_points']
        results['GGA_mask_valid'] = a['GGA_mask2d'] & a['GGA_mask_valid'] & a['GGA_mask_depth']
        results['GGA_bdry_masks'] = a['GGA_bdry_masks']
        results['GGA_difficulty'] = a['difficulty']
        results['GGA_num_points_in_box2d'] = a['GGA_num_points_in_box2d']
        return results

    def __call__(self, results):
        if self.with_bbox:          # mmdet LoadAnnotations._load_bboxes for plain arrays
            results['gt_bboxes'] = results['ann_info']['bboxes'].copy()
            results['bbox_fields'].append('gt_bboxes')
        if self.with_label:
            results['gt_labels'] = results['ann_info']['labels'].copy()
        if self.with_bbox_3d:
            results = self._load_bboxes_3d(results)
        if self.with_label_3d:
            results = self._load_labels_3d(results)
        if self.with_gga:
            results = self._load_GGA_labels(results)
        return results


def lidar_boxes_to_camera(boxes, rt_mat):
    """``Box3DMode.convert(box, LIDAR, CAM, rt_mat)`` for an [N,7] tensor (box_3d_mode.py:113-199): sizes (x, y, z) ->
    (x, z, y), yaw -> limit_period(-yaw - pi/2, 2 pi), centres through ``rt_mat`` (cast to the boxes' dtype)."""
    arr = boxes.clone()
    xyz_size = torch.cat([arr[..., 3:4], arr[..., 5:6], arr[..., 4:5]], dim=-1)
    yaw = limit_period(-arr[..., 6:7] - np.pi / 2, period=np.pi * 2)
    rt = arr.new_tensor(np.asarray(rt_mat))
    if rt.size(1) == 4:
        xyz = torch.cat([arr[..., :3], arr.new_ones(arr.size(0), 1)], dim=-1) @ rt.t()
    else:
        xyz = arr[..., :3] @ rt.t()
    return torch.cat([xyz[..., :3], xyz_size, yaw, arr[..., 7:]], dim=-1)


@DATASETS.register_module()
class KittiDataset_GGA_match(KittiDataset_GGA_train):
    """The dataset of configs/gga/gga_kitti_matching_config.py (mmdet3d/datasets/kitti_dataset_GGA_match.py): the train dataset
    whose ``evaluate`` turns the detections of a test run into KITTI annotations (``format_results`` ->
    ``bbox2result_kitti`` -> ``convert_valid_bboxes``, :330-383,458-571,685-766) and hands them to
    ``pseudo_label_matching_kitti`` together with a copy of its infos (:419-424) - the step that writes the pseudo-label
    file of the GGA recipe. The KITTI AP evaluation that follows in the reference (``kitti_eval``, a numba CPU code of
    mmdet3d/core/evaluation) is out of scope (evaluation is not on the path); ``evaluate`` returns the counts of
    the matching instead."""

    def format_results(self, outputs, pklfile_prefix=None, submission_prefix=None):
        import tempfile
        tmp_dir = None
        if pklfile_prefix is None:
            tmp_dir = tempfile.TemporaryDirectory()
            pklfile_prefix = os.path.join(tmp_dir.name, 'results')
        if not isinstance(outputs[0], dict):
            raise NotImplementedError('2D-only results (bbox2result_kitti2d) are not produced by the GGA configs')
        if 'pts_bbox' in outputs[0] or 'img_bbox' in outputs[0]:
            result_files = dict()
            for name in outputs[0]:
                if 'img' in name:
                    raise NotImplementedError('image-branch results are not produced by the GGA configs')
                results_ = [out[name] for out in outputs]
                sub = submission_prefix + name if submission_prefix is not None else None
                result_files[name] = self.bbox2result_kitti(results_, self.CLASSES, pklfile_prefix + name, sub)
        else:
            result_files = self.bbox2result_kitti(outputs, self.CLASSES, pklfile_prefix, submission_prefix)
        return result_files, tmp_dir

    def evaluate(self, results, metric=None, logger=None, pklfile_prefix=None, submission_prefix=None, show=False, out_dir=None,
                 pipeline=None, pseudo_label_file='default', device='cuda:0'):
        from .pseudo_labels import DEFAULT_OUT_FILE, pseudo_label_matching_kitti
        result_files, tmp_dir = self.format_results(results, pklfile_prefix)
        dets = result_files['pts_bbox'] if isinstance(result_files, dict) else result_files
        infos = copy.deepcopy(self.data_infos)
        out_file = DEFAULT_OUT_FILE if pseudo_label_file == 'default' else pseudo_label_file
        gt_annos = pseudo_label_matching_kitti(infos, dets, filename=out_file, device=device)
        if tmp_dir is not None:
            tmp_dir.cleanup()
        return {'pseudo_labels/frames': float(len(gt_annos)), 'pseudo_labels/objects': float(sum(len(a['name']) for a in gt_annos)),
                'pseudo_labels/detections': float(sum(len(d['name']) for d in dets))}

    def bbox2result_kitti(self, net_outputs, class_names, pklfile_prefix=None, submission_prefix=None):
        assert len(net_outputs) == len(self.data_infos), 'invalid list length of network outputs'
        if submission_prefix is not None:
            os.makedirs(submission_prefix, exist_ok=True)
        det_annos = []
        for idx, pred_dicts in enumerate(net_outputs):
            info = self.data_infos[idx]
            sample_idx = info['image']['image_idx']
            image_shape = info['image']['image_shape'][:2]
            box_dict = self.convert_valid_bboxes(pred_dicts, info)
            n = len(box_dict['bbox'])
            cam, lidar = np.asarray(box_dict['box3d_camera']), np.asarray(box_dict['box3d_lidar'])
            # KITTI label columns of the frame's valid detections, whole columns at a time: 2D box clipped to the image,
            # observation angle alpha = rotation_y minus the azimuth of the box centre seen from the LiDAR origin
            bbox = np.array(box_dict['bbox'], dtype=np.asarray(box_dict['bbox']).dtype).reshape(n, 4)
            if n:
                bbox[:, 2:] = np.minimum(bbox[:, 2:], image_shape[::-1])
                bbox[:, :2] = np.maximum(bbox[:, :2], [0, 0])
                anno = dict(name=np.array([class_names[int(label)] for label in box_dict['label_preds']]),
                            truncated=np.zeros(n), occluded=np.zeros(n, dtype=np.int64),
                            alpha=-np.arctan2(-lidar[:, 1], lidar[:, 0]) + cam[:, 6], bbox=bbox, dimensions=cam[:, 3:6],
                            location=cam[:, :3], rotation_y=cam[:, 6], score=np.asarray(box_dict['scores']))
            else:
                anno = dict(name=np.array([]), truncated=np.array([]), occluded=np.array([]), alpha=np.array([]), bbox=np.zeros([0, 4]),
                            dimensions=np.zeros([0, 3]), location=np.zeros([0, 3]), rotation_y=np.array([]), score=np.array([]))
            if submission_prefix is not None:
                with open(f'{submission_prefix}/{sample_idx:06d}.txt', 'w') as f:
                    bbox, loc, dims = anno['bbox'], anno['location'], anno['dimensions']          # lhw -> hwl
                    for i in range(len(bbox)):
                        print('{} -1 -1 {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f}'.format(
                            anno['name'][i], anno['alpha'][i], bbox[i][0], bbox[i][1], bbox[i][2], bbox[i][3], dims[i][1], dims[i][2],
                            dims[i][0], loc[i][0], loc[i][1], loc[i][2], anno['rotation_y'][i], anno['score'][i]), file=f)
            anno['sample_idx'] = np.array([sample_idx] * len(anno['score']), dtype=np.int64)
            det_annos.append(anno)
        if pklfile_prefix is not None:
            out = pklfile_prefix if pklfile_prefix.endswith(('.pkl', '.pickle')) else f'{pklfile_prefix}.pkl'
            with open(out, 'wb') as f:
                pickle.dump(det_annos, f)
        return det_annos

    def convert_valid_bboxes(self, box_dict, info):
        from .box3d import CameraInstance3DBoxes, points_cam2img
        box_preds, scores, labels = box_dict['boxes_3d'], box_dict['scores_3d'], box_dict['labels_3d']
        sample_idx = info['image']['image_idx']
        box_preds.limit_yaw(offset=0.5, period=np.pi * 2)
        empty = dict(bbox=np.zeros([0, 4]), box3d_camera=np.zeros([0, 7]), box3d_lidar=np.zeros([0, 7]), scores=np.zeros([0]),
                     label_preds=np.zeros([0, 4]), sample_idx=sample_idx)
        if len(box_preds) == 0:
            return empty
        rect = info['calib']['R0_rect'].astype(np.float32)
        Trv2c = info['calib']['Tr_velo_to_cam'].astype(np.float32)
        P2 = box_preds.tensor.new_tensor(info['calib']['P2'].astype(np.float32))
        img_shape = info['image']['image_shape']
        cam = CameraInstance3DBoxes(lidar_boxes_to_camera(box_preds.tensor, rect @ Trv2c), box_dim=box_preds.tensor.shape[-1])
        box_corners_in_image = points_cam2img(cam.corners, P2)
        minxy, maxxy = torch.min(box_corners_in_image, dim=1)[0], torch.max(box_corners_in_image, dim=1)[0]
        box_2d_preds = torch.cat([minxy, maxxy], dim=1)
        image_shape = box_preds.tensor.new_tensor(img_shape)
        valid_cam_inds = ((box_2d_preds[:, 0] < image_shape[1]) & (box_2d_preds[:, 1] < image_shape[0]) &
                          (box_2d_preds[:, 2] > 0) & (box_2d_preds[:, 3] > 0))
        limit_range = box_preds.tensor.new_tensor(self.pcd_limit_range)
        center = box_preds.tensor[:, :3]              # ``center`` of a LiDAR box is its bottom centre (base_box3d.py:96-103)
        valid_inds = valid_cam_inds & ((center > limit_range[:3]) & (center < limit_range[3:])).all(-1)
        if valid_inds.sum() > 0:
            return dict(bbox=box_2d_preds[valid_inds, :].numpy(), box3d_camera=cam.tensor[valid_inds].numpy(),
                        box3d_lidar=box_preds.tensor[valid_inds].numpy(), scores=scores[valid_inds].numpy(),
                        label_preds=labels[valid_inds].numpy(), sample_idx=sample_idx)
        return empty
