"""The GGA train data pipeline — SURVEY.md §8(f) rank 2, the step before the hot path.

Host-side mirror (Python, as in the reference) of the transforms named by
``train_pipeline`` in ``configs/gga/gga_kitti_config.py:111-137`` of the reference:

* ``ObjectSample_GGA`` + ``DataBaseSampler_GGA`` + ``BatchSampler``
  (mmdet3d/datasets/pipelines/gga_processing.py:21-198, 588-1011): ground-truth database sampling
  with the centre-distance collision test (``min_distance``), removal of scene points around the
  pasted objects, concatenation of the GGA side information;
* ``PointsRangeFilter`` / ``PointShuffle`` (mmdet3d/datasets/pipelines/transforms_3d.py:942-977,
  858-883);
* ``ObjectRangeFilter_GGA`` (gga_processing.py:213-310);
* ``DefaultFormatBundle3D_GGA`` / ``Collect3D_GGA`` (gga_processing.py:384-583) for the keys the
  GGA detector consumes;
* ``Compose`` and a minimal ``DataContainer`` (mmcv.parallel, third-party).

Random draws are made with the same generators in the same order as the reference
(``np.random.shuffle`` in ``BatchSampler``, ``torch.randperm`` in ``PointShuffle``), so with equal
seeds the augmentation stream is identical — that is what ``tests/test_pipelines.py`` pins against
vectors produced by the reference's classes.

The point-level part (remove scene points near pasted objects, concatenate, range filter, shuffle)
also exists as ONE batched device op, ``functional.points_prepare_batch`` (``gga_points_prepare_batch``
in include/gga_hip.h): ``DevicePointPrep`` below collects what it needs from samples whose
``ObjectSample_GGA`` / ``PointsRangeFilter`` / ``PointShuffle`` ran with ``defer_points=True`` and
hands the voxelizer device-resident frames with device-side counts (no host round trip).
"""
import copy
import os
import pickle

import numpy as np
import torch
from scipy.spatial.distance import cdist, pdist, squareform

from .box3d import LiDARInstance3DBoxes
from .points import BasePoints, LiDARPoints
from .registry import OBJECTSAMPLERS, PIPELINES, build_from_cfg


# ----------------------------------------------------------------------------- containers
class DataContainer:
    """mmcv.parallel.DataContainer: a tagged payload for the collate function."""

    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        self._data, self._stack, self._padding_value = data, stack, padding_value
        self._cpu_only, self._pad_dims = cpu_only, pad_dims

    data = property(lambda self: self._data)
    stack = property(lambda self: self._stack)
    padding_value = property(lambda self: self._padding_value)
    cpu_only = property(lambda self: self._cpu_only)
    pad_dims = property(lambda self: self._pad_dims)
    datatype = property(lambda self: self._data.type() if isinstance(self._data, torch.Tensor) else type(self._data))

    def __repr__(self):
        return f'{self.__class__.__name__}({self._data!r})'


DC = DataContainer


def to_tensor(data):
    """mmdet.datasets.pipelines.to_tensor."""
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, np.ndarray):
        return torch.from_numpy(data)
    if isinstance(data, (list, tuple)) and not isinstance(data, str):
        return torch.tensor(data)
    if isinstance(data, int):
        return torch.LongTensor([data])
    if isinstance(data, float):
        return torch.FloatTensor([data])
    raise TypeError(f'type {type(data)} cannot be converted to tensor.')


@PIPELINES.register_module()
class Compose:
    def __init__(self, transforms):
        self.transforms = [build_from_cfg(t, PIPELINES) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


@PIPELINES.register_module()
class LoadPointsFromFile:
    """transforms: mmdet3d/datasets/pipelines/loading.py LoadPointsFromFile for local ``.bin`` files
    (float32 rows of ``load_dim`` values) — what the database sampler's ``points_loader`` needs."""

    def __init__(self, coord_type='LIDAR', load_dim=6, use_dim=[0, 1, 2], shift_height=False, use_color=False,
                 file_client_args=dict(backend='disk')):
        assert coord_type == 'LIDAR' and not shift_height and not use_color
        self.load_dim = load_dim
        self.use_dim = list(range(use_dim)) if isinstance(use_dim, int) else list(use_dim)
        assert max(self.use_dim) < load_dim

    def __call__(self, results):
        pts = np.fromfile(results['pts_filename'], dtype=np.float32).reshape(-1, self.load_dim)[:, self.use_dim]
        results['points'] = LiDARPoints(pts, points_dim=pts.shape[-1], attribute_dims=None)
        return results


# ----------------------------------------------------------------------------- database sampling
class BatchSampler:
    """A deck of database entries (gga_processing.py:588-654, same constructor and ``sample``): entries are dealt in a
    shuffled order, ``num`` at a time; a request that reaches the end of the deck gets only what is left (possibly fewer
    than asked) and the deck is shuffled again. The calls to ``np.random.shuffle`` - one at construction, one per new pass -
    are part of the behaviour: the reference's random stream is reproduced draw for draw (tests/test_pipelines.py)."""

    def __init__(self, sampled_list, name=None, epoch=None, shuffle=True, drop_reminder=False):
        self.entries, self.name, self.shuffled = sampled_list, name, shuffle
        self.deck = np.arange(len(sampled_list))
        self.dealt = 0
        if shuffle:
            np.random.shuffle(self.deck)

    def _next_pass(self):
        assert self.name is not None
        if self.shuffled:
            np.random.shuffle(self.deck)
        self.dealt = 0

    def sample(self, num):
        upto = self.dealt + num
        if upto >= len(self.entries):
            hand = self.deck[self.dealt:].copy()
            self._next_pass()
        else:
            hand, self.dealt = self.deck[self.dealt:upto], upto
        return [self.entries[i] for i in hand]


def collision_free(anchor_xy, cand_xy, min_distance):
    """The collision test of ``sample_class_GGA`` (gga_processing.py:984-1010): candidates are
    visited in order; one closer than ``min_distance`` (BEV centre distance, scipy ``pdist``) to an
    anchor or to any candidate that has not been dropped yet - accepted earlier ones AND all later
    ones - is dropped, which removes it from the later tests. -> bool [n_cand]."""
    n_a, n_c = len(anchor_xy), len(cand_xy)
    total = np.concatenate([np.asarray(anchor_xy, np.float64).reshape(-1, 2),
                            np.asarray(cand_xy, np.float64).reshape(-1, 2)], axis=0)
    coll = squareform(pdist(total)) < min_distance
    coll[:n_a, :n_a] = False
    coll[np.arange(n_a + n_c), np.arange(n_a + n_c)] = False
    ok = np.zeros(n_c, bool)
    for i in range(n_a, n_a + n_c):
        if coll[i].any():
            coll[i] = False
            coll[:, i] = False
        else:
            ok[i - n_a] = True
    return ok


_SAMPLED_FIELDS = (          # key of the returned dict, key inside a database record
    ('gt_bbox_3ds', 'box3d_lidar'), ('GGA_box_imgs', 'GGA_box_img'), ('GGA_lidar2imgs', 'GGA_lidar2img'),
    ('GGA_init_pseudo_labels', 'GGA_init_pseudo_label'), ('GGA_bdry_masks', 'GGA_bdry_mask'),
    ('GGA_difficulties', 'difficulty'), ('GGA_num_points_in_box2ds', 'GGA_num_points_in_box2d'))


@OBJECTSAMPLERS.register_module()
class DataBaseSampler_GGA:
    """gga_processing.py:656-1011. ``info_path``: pickle of ``{class: [record, ...]}`` (or the dict
    itself); every record carries ``name, path, box3d_lidar, difficulty, num_points_in_gt`` and the
    GGA fields (``GGA_init_pseudo_label, GGA_box_img, GGA_lidar2img, GGA_bdry_mask, GGA_mask2d,
    GGA_mask_depth, GGA_mask_valid, GGA_num_points_in_box2d, GGA_in_box_points``)."""

    def __init__(self, info_path, data_root, rate, prepare, sample_groups, classes=None, bbox_code_size=None,
                 points_loader=dict(type='LoadPointsFromFile', coord_type='LIDAR', load_dim=4, use_dim=[0, 1, 2, 3]),
                 file_client_args=dict(backend='disk')):
        self.data_root, self.info_path, self.rate, self.prepare, self.classes = data_root, info_path, rate, prepare, classes
        self.cat2label = {name: i for i, name in enumerate(classes)}
        self.label2cat = {i: name for i, name in enumerate(classes)}
        self.points_loader = build_from_cfg(points_loader, PIPELINES) if isinstance(points_loader, dict) else points_loader
        if isinstance(info_path, dict):
            db_infos = info_path
        else:
            with open(info_path, 'rb') as f:
                db_infos = pickle.load(f)
        for prep_func, val in prepare.items():
            db_infos = getattr(self, prep_func)(db_infos, val)
        self.db_infos = db_infos
        self.bbox_code_size = bbox_code_size
        if bbox_code_size is not None:
            for infos in self.db_infos.values():
                for info in infos:
                    info['box3d_lidar'] = info['box3d_lidar'][:bbox_code_size]
        self.sample_groups = [{name: int(num)} for name, num in sample_groups.items()]
        self.group_db_infos = self.db_infos
        self.sample_classes = [k for g in self.sample_groups for k in g.keys()]
        self.sample_max_nums = [v for g in self.sample_groups for v in g.values()]
        self.sampler_dict = {k: BatchSampler(v, k, shuffle=True) for k, v in self.group_db_infos.items()}

    @staticmethod
    def filter_by_difficulty(db_infos, removed_difficulty):
        return {k: [i for i in v if i['difficulty'] not in removed_difficulty] for k, v in db_infos.items()}

    @staticmethod
    def filter_by_min_points(db_infos, min_gt_points_dict):
        for name, min_num in min_gt_points_dict.items():
            if int(min_num) > 0:
                db_infos[name] = [i for i in db_infos[name] if i['num_points_in_gt'] >= int(min_num)]
        return db_infos

    def sample_class_GGA(self, name, num, est_points_mean, min_distance):
        """Draw ``num`` records of class ``name``, keep the valid ones that pass the collision test
        against ``est_points_mean`` (existing + already pasted object centres)."""
        # (The reference deep-copies the drawn records here, gga_processing.py:974 - 3.7 of the 11 ms this pipeline takes per
        # frame. Nothing downstream writes into a record or its arrays: every field is stacked / concatenated into new arrays
        # and the in-box point sets are only read, so the records are shared.)
        sampled = list(self.sampler_dict[name].sample(num))
        valid = np.stack([s['GGA_mask_valid'] for s in sampled], axis=0)
        sampled = [sampled[i] for i in np.arange(len(sampled))[valid]]
        if not sampled:
            return []
        cand = np.stack([s['GGA_init_pseudo_label'][:2] for s in sampled], axis=0)
        ok = collision_free(est_points_mean[:, :2], cand, min_distance)
        return [s for s, keep in zip(sampled, ok) if keep]

    def sample_all(self, GGA_init_pseudo_labels, gt_labels, mask_valid, min_distance=5.0, ground_plane=None):
        avoid = GGA_init_pseudo_labels[mask_valid]
        nums = []
        for cname, max_num in zip(self.sample_classes, self.sample_max_nums):
            label = self.cat2label[cname]
            n = int(max_num - np.sum([g == label for g in gt_labels]))
            nums.append(np.round(self.rate * n).astype(np.int64))

        sampled = []
        cols = {k: [] for k, _ in _SAMPLED_FIELDS}
        cols['GGA_mask_valids'] = []
        in_box = []
        for cname, n in zip(self.sample_classes, nums):
            if n <= 0:
                continue
            got = self.sample_class_GGA(cname, n, avoid, min_distance)
            sampled += got
            if not got:
                continue
            for key, rec in _SAMPLED_FIELDS:
                cols[key].append(np.stack([s[rec] for s in got], axis=0))
            cols['GGA_mask_valids'].append(np.stack([s['GGA_mask2d'] & s['GGA_mask_depth'] & s['GGA_mask_valid']
                                                     for s in got], axis=0))
            in_box += [s['GGA_in_box_points'] for s in got]
            avoid = np.concatenate([avoid, cols['GGA_init_pseudo_labels'][-1]], axis=0)
        if not sampled:
            return None
        ret = {k: np.concatenate(v, axis=0) for k, v in cols.items()}
        pts = []
        for info in sampled:
            path = os.path.join(self.data_root, info['path']) if self.data_root else info['path']
            pts.append(self.points_loader(dict(pts_filename=path))['points'])
        ret['gt_labels_3d'] = np.array([self.cat2label[s['name']] for s in sampled], dtype=np.int64)
        ret['GGA_in_box_points'] = in_box
        ret['points'] = pts[0].cat(pts)
        ret['group_ids'] = np.arange(mask_valid.shape[0], mask_valid.shape[0] + len(sampled))
        return ret


_GGA_OBJECT_KEYS = ('GGA_boxes_img', 'GGA_lidar2img', 'GGA_init_pseudo_labels', 'GGA_mask_valid', 'GGA_bdry_masks',
                    'GGA_difficulty', 'GGA_num_points_in_box2d')
_GGA_SAMPLED_KEYS = ('GGA_box_imgs', 'GGA_lidar2imgs', 'GGA_init_pseudo_labels', 'GGA_mask_valids', 'GGA_bdry_masks',
                     'GGA_difficulties', 'GGA_num_points_in_box2ds')


@PIPELINES.register_module()
class ObjectSample_GGA:
    """gga_processing.py:21-211. ``defer_points=True`` (extension) leaves ``points`` untouched and
    records ``sampled_points`` / ``sampled_centers`` for the batched device op instead."""

    def __init__(self, min_distance, db_sampler, sample_2d=False, use_ground_plane=False, defer_points=False):
        assert not sample_2d, 'image pasting is not part of the GGA LiDAR recipe'
        self.sampler_cfg = db_sampler
        self.min_distance = min_distance
        self.sample_2d = sample_2d
        if isinstance(db_sampler, dict):
            if 'type' not in db_sampler:
                db_sampler = dict(db_sampler, type='DataBaseSampler_GGA')
            db_sampler = build_from_cfg(db_sampler, OBJECTSAMPLERS)
        self.db_sampler = db_sampler
        self.use_ground_plane = use_ground_plane
        self.defer_points = defer_points

    @staticmethod
    def remove_points_in_boxes_v2(points, pts_mean, min_distance):
        """Drop the points whose BEV distance (float64, scipy ``cdist``) to any pasted object's
        centre is below ``min_distance`` — there are no 3D boxes to test against."""
        near = cdist(points.tensor.numpy()[:, :2], pts_mean[:, :2]) < min_distance
        return points[np.logical_not(near.any(-1))]

    def __call__(self, d):
        boxes, labels, in_box = d['gt_bboxes_3d'], d['gt_labels_3d'], d['GGA_in_box_points']
        gga = {k: d[k] for k in _GGA_OBJECT_KEYS}
        points = d['points']
        got = self.db_sampler.sample_all(gga['GGA_init_pseudo_labels'], labels, gga['GGA_mask_valid'], self.min_distance,
                                         ground_plane=None)
        if got is not None:
            labels = np.concatenate([labels, got['gt_labels_3d']], axis=0)
            boxes = boxes.new_box(np.concatenate([boxes.tensor.numpy(), got['gt_bbox_3ds']]))
            for k, ks in zip(_GGA_OBJECT_KEYS, _GGA_SAMPLED_KEYS):
                gga[k] = np.concatenate([gga[k], got[ks]])
            in_box += got['GGA_in_box_points']
            centers = got['GGA_init_pseudo_labels'][:, :2]
            if self.defer_points:
                d['sampled_points'], d['sampled_centers'] = got['points'], np.ascontiguousarray(centers, np.float64)
            else:
                points = self.remove_points_in_boxes_v2(points, centers, self.min_distance)
                points = points.cat([got['points'], points])
        d['points'], d['gt_bboxes_3d'], d['gt_labels_3d'] = points, boxes, labels.astype(np.int64)
        d.update(gga)
        d['GGA_in_box_points'] = in_box
        return d


# ----------------------------------------------------------------------------- filters
@PIPELINES.register_module()
class PointsRangeFilter:
    """transforms_3d.py:942-977. ``defer_points=True``: only record the range (device op does it)."""

    def __init__(self, point_cloud_range, defer_points=False):
        self.pcd_range = np.array(point_cloud_range, dtype=np.float32)
        self.defer_points = defer_points

    def __call__(self, d):
        if self.defer_points:
            d['deferred_point_range'] = self.pcd_range
            return d
        points = d['points']
        mask = points.in_range_3d(self.pcd_range)
        d['points'] = points[mask]
        mask = mask.numpy()
        for k in ('pts_instance_mask', 'pts_semantic_mask'):
            if d.get(k) is not None:
                d[k] = d[k][mask]
        return d


@PIPELINES.register_module()
class PointShuffle:
    """transforms_3d.py:858-883 (``torch.randperm``). ``defer_points=True``: draw a 63-bit seed from
    the same torch generator instead; the device op permutes with it."""

    def __init__(self, defer_points=False):
        self.defer_points = defer_points

    def __call__(self, d):
        if self.defer_points:
            d['deferred_shuffle_seed'] = int(torch.randint(1, 2 ** 62, (1,)).item())
            return d
        idx = d['points'].shuffle().numpy()
        for k in ('pts_instance_mask', 'pts_semantic_mask'):
            if d.get(k) is not None:
                d[k] = d[k][idx]
        return d


@PIPELINES.register_module()
class ObjectRangeFilter_GGA:
    """gga_processing.py:213-310: keep objects that are valid, have more than ``num_points_range``
    points in their 2D box, a known difficulty and a pseudo-label centre inside the BEV range."""

    def __init__(self, point_cloud_range, num_points_range):
        self.pcd_range = np.array(point_cloud_range, dtype=np.float32)
        self.num_points_range = num_points_range

    def __call__(self, d):
        assert isinstance(d['gt_bboxes_3d'], LiDARInstance3DBoxes)
        lo_x, lo_y, hi_x, hi_y = self.pcd_range[[0, 1, 3, 4]]
        c = d['GGA_init_pseudo_labels'][:, :2]
        in_range = (c[:, 0] > lo_x) & (c[:, 1] > lo_y) & (c[:, 0] < hi_x) & (c[:, 1] < hi_y)
        mask = d['GGA_mask_valid'] & (d['GGA_num_points_in_box2d'] > self.num_points_range) & (d['GGA_difficulty'] > -1) & in_range
        d['gt_labels_3d'] = d['gt_labels_3d'][mask]
        boxes = d['gt_bboxes_3d'][mask]
        boxes.limit_yaw(offset=0.5, period=2 * np.pi)
        d['gt_bboxes_3d'] = boxes
        for k in ('GGA_boxes_img', 'GGA_bdry_masks', 'GGA_lidar2img', 'GGA_init_pseudo_labels'):
            d[k] = d[k][mask]
        d['GGA_in_box_points'] = [p for p, keep in zip(d['GGA_in_box_points'], mask.tolist()) if keep]
        return d


# ----------------------------------------------------------------------------- formatting
@PIPELINES.register_module()
class DefaultFormatBundle3D_GGA:
    """gga_processing.py:384-492 for the LiDAR / GGA keys."""

    def __init__(self, class_names, with_gt=True, with_label=True):
        self.class_names, self.with_gt, self.with_label = class_names, with_gt, with_label

    def __call__(self, d):
        if 'points' in d:
            assert isinstance(d['points'], BasePoints)
            d['points'] = DC(d['points'].tensor)
        if self.with_gt and self.with_label:
            if 'gt_names_3d' in d:
                d['gt_labels_3d'] = np.array([self.class_names.index(n) for n in d['gt_names_3d']], dtype=np.int64)
            for k in ('GGA_boxes_img', 'GGA_lidar2img', 'GGA_bdry_masks', 'GGA_init_pseudo_labels'):
                if k in d:
                    d[k] = DC(torch.from_numpy(d[k]))
            if isinstance(d.get('GGA_in_box_points'), list):
                d['GGA_in_box_points'] = DC([to_tensor(p) for p in d['GGA_in_box_points']])
        for k in ('gt_labels_3d',):
            if k in d:
                d[k] = DC(to_tensor(d[k]))
        if 'gt_bboxes_3d' in d:
            b = d['gt_bboxes_3d']
            d['gt_bboxes_3d'] = DC(b, cpu_only=True) if isinstance(b, LiDARInstance3DBoxes) else DC(to_tensor(b))
        return d


@PIPELINES.register_module()
class Collect3D_GGA:
    """gga_processing.py:494-586: ``keys`` + an ``img_metas`` container with the meta keys present."""

    META_KEYS = ('filename', 'ori_shape', 'img_shape', 'lidar2img', 'depth2img', 'cam2img', 'pad_shape', 'scale_factor',
                 'flip', 'pcd_horizontal_flip', 'pcd_vertical_flip', 'box_mode_3d', 'box_type_3d', 'img_norm_cfg',
                 'pcd_trans', 'sample_idx', 'pcd_scale_factor', 'pcd_rotation', 'pcd_rotation_angle', 'pts_filename',
                 'transformation_3d_flow', 'trans_mat', 'affine_aug')

    def __init__(self, keys, meta_keys=META_KEYS):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, d):
        out = {'img_metas': DC({k: d[k] for k in self.meta_keys if k in d}, cpu_only=True)}
        for k in self.keys:
            out[k] = d[k]
        for k in ('sampled_points', 'sampled_centers', 'deferred_point_range', 'deferred_shuffle_seed'):
            if k in d:          # hand-over to DevicePointPrep
                out[k] = d[k]
        return out


# ----------------------------------------------------------------------------- device hand-over
class DevicePointPrep:
    """Batched, device-side tail of the point pipeline for samples produced with
    ``defer_points=True``: uploads the raw scene points (+ the pasted objects' points and centres)
    of a whole batch once and runs remove-near-centres + concatenate + range filter + shuffle as
    one ``gga_points_prepare_batch`` call. Returns what ``Voxelization.forward_prepared`` takes."""

    def __init__(self, min_distance, device='cuda:0'):
        self.min_distance, self.device = float(min_distance), torch.device(device)

    def __call__(self, samples):
        from . import functional as F
        unwrap = lambda v: v.data if isinstance(v, DataContainer) else v
        scene = [unwrap(s['points']) for s in samples]
        scene = [p.tensor if isinstance(p, BasePoints) else p for p in scene]
        sampled = [s['sampled_points'].tensor if 'sampled_points' in s else scene[0].new_zeros((0, scene[0].shape[1]))
                   for s in samples]
        centers = [s.get('sampled_centers', np.zeros((0, 2))) for s in samples]
        rng = samples[0].get('deferred_point_range')
        assert rng is not None, 'PointsRangeFilter(defer_points=True) must be part of the pipeline'
        seeds = [int(s.get('deferred_shuffle_seed', 0)) for s in samples]
        return F.points_prepare_batch(scene, sampled, centers, self.min_distance, rng, seeds, self.device)


# ----------------------------------------------------------------------------- test-time pipeline (pseudo-label run)
# The transforms of the reference's test_pipeline (configs/gga/gga_kitti_config.py:139-163: MultiScaleFlipAug3D around
# GlobalRotScaleTrans / RandomFlip3D / PointsRangeFilter / DefaultFormatBundle3D / Collect3D; test_time_aug.py:118-229,
# transforms_3d.py:78-211,544-724, formating.py) for LiDAR points. With the parameters of the GGA configs they change nothing
# (one scale, no flip, zero rotation / translation); the point-side arithmetic is there for other settings, box fields are
# only handled where no box has to move.
def _rotate_points_z(points, angle):
    """Row vectors times [[c, s, 0], [-s, c, 0], [0, 0, 1]] (rotation_3d_in_axis about z, structures/utils.py:79-117)."""
    c, s = float(np.cos(angle)), float(np.sin(angle))
    rot_t = points.tensor.new_tensor([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])
    points.tensor[:, :3] = points.tensor[:, :3] @ rot_t
    return rot_t


def _no_boxes_to_move(d, what):
    for field in d.get('bbox3d_fields', []):
        if len(d[field]):
            raise NotImplementedError(f'{what} of 3D boxes is not on the GGA path (its train pipeline has no such transform)')


@PIPELINES.register_module()
class GlobalRotScaleTrans:
    def __init__(self, rot_range=[-0.78539816, 0.78539816], scale_ratio_range=[0.95, 1.05], translation_std=[0, 0, 0], shift_height=False):
        as_pair = lambda v: [-v, v] if isinstance(v, (int, float)) else list(v)
        self.rot_range, self.scale_ratio_range = as_pair(rot_range), list(scale_ratio_range)
        self.translation_std = [translation_std] * 3 if isinstance(translation_std, (int, float)) else list(translation_std)
        assert all(t >= 0 for t in self.translation_std) and not shift_height

    def __call__(self, d):
        d.setdefault('transformation_3d_flow', [])
        angle = np.random.uniform(self.rot_range[0], self.rot_range[1])          # the draws come in this order: rotation,
        if angle != 0:
            _no_boxes_to_move(d, 'rotation')
        d['pcd_rotation'] = _rotate_points_z(d['points'], angle)
        d['pcd_rotation_angle'] = angle
        if 'pcd_scale_factor' not in d:                                        # scale (unless the test-time wrapper fixed it),
            d['pcd_scale_factor'] = np.random.uniform(self.scale_ratio_range[0], self.scale_ratio_range[1])
        if d['pcd_scale_factor'] != 1:
            _no_boxes_to_move(d, 'scaling')
            d['points'].tensor[:, :3] *= d['pcd_scale_factor']
        shift = np.random.normal(scale=np.array(self.translation_std, dtype=np.float32), size=3).T      # translation
        if np.any(shift != 0):
            _no_boxes_to_move(d, 'translation')
            d['points'].tensor[:, :3] += d['points'].tensor.new_tensor(shift)
        d['pcd_trans'] = shift
        d['transformation_3d_flow'].extend(['R', 'S', 'T'])
        return d


@PIPELINES.register_module()
class RandomFlip3D:
    """Flips decided upstream (``flip`` / ``pcd_horizontal_flip`` / ``pcd_vertical_flip`` set by MultiScaleFlipAug3D) or drawn
    with the given ratios; LiDAR points: horizontal = y -> -y, vertical = x -> -x."""

    def __init__(self, sync_2d=True, flip_ratio_bev_horizontal=0.0, flip_ratio_bev_vertical=0.0, **kwargs):
        self.sync_2d, self.h_ratio, self.v_ratio = sync_2d, flip_ratio_bev_horizontal, flip_ratio_bev_vertical

    def __call__(self, d):
        d.setdefault('flip', False)
        if self.sync_2d:
            d['pcd_horizontal_flip'], d['pcd_vertical_flip'] = d['flip'], False
        else:
            if 'pcd_horizontal_flip' not in d:
                d['pcd_horizontal_flip'] = bool(np.random.rand() < self.h_ratio)
            if 'pcd_vertical_flip' not in d:
                d['pcd_vertical_flip'] = bool(np.random.rand() < self.v_ratio)
        d.setdefault('transformation_3d_flow', [])
        for flag, axis, tag in (('pcd_horizontal_flip', 1, 'HF'), ('pcd_vertical_flip', 0, 'VF')):
            if d[flag]:
                _no_boxes_to_move(d, 'flipping')
                d['points'].tensor[:, axis] = -d['points'].tensor[:, axis]
                d['transformation_3d_flow'].append(tag)
        return d


@PIPELINES.register_module()
class MultiScaleFlipAug3D:
    """Test-time augmentation wrapper: the inner transforms run once per (image scale, point scale, flip) combination on a deep
    copy of the sample, and the results are regrouped key by key into lists - one entry per combination (exactly one for the
    GGA configs)."""

    def __init__(self, transforms, img_scale, pts_scale_ratio, flip=False, flip_direction='horizontal', pcd_horizontal_flip=False,
                 pcd_vertical_flip=False):
        self.transforms = Compose(transforms)
        self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
        self.pts_scale_ratio = [float(r) for r in (pts_scale_ratio if isinstance(pts_scale_ratio, list) else [pts_scale_ratio])]
        self.flip, self.h_flip, self.v_flip = flip, pcd_horizontal_flip, pcd_vertical_flip
        self.flip_direction = flip_direction if isinstance(flip_direction, list) else [flip_direction]

    def __call__(self, d):
        import copy
        import itertools
        hs = [False, True] if self.flip and self.h_flip else [False]
        vs = [False, True] if self.flip and self.v_flip else [False]
        runs = []
        for scale, ratio, h, v, direction in itertools.product(self.img_scale, self.pts_scale_ratio, hs, vs, self.flip_direction):
            one = copy.deepcopy(d)
            one.update(scale=scale, flip=bool(self.flip), pcd_scale_factor=ratio, flip_direction=direction, pcd_horizontal_flip=h,
                       pcd_vertical_flip=v)
            runs.append(self.transforms(one))
        return {key: [r[key] for r in runs] for key in runs[0]}


@PIPELINES.register_module()
class DefaultFormatBundle3D:
    """formating.py ``DefaultFormatBundle3D`` for the LiDAR keys: points as a tensor container; with labels, the label / box
    containers of ``DefaultFormatBundle3D_GGA`` minus the GGA side arrays."""

    def __init__(self, class_names, with_gt=True, with_label=True):
        self.class_names, self.with_gt, self.with_label = class_names, with_gt, with_label

    def __call__(self, d):
        if 'points' in d:
            assert isinstance(d['points'], BasePoints)
            d['points'] = DC(d['points'].tensor)
        if self.with_gt and self.with_label:
            if 'gt_names_3d' in d:
                d['gt_labels_3d'] = np.array([self.class_names.index(n) for n in d['gt_names_3d']], dtype=np.int64)
            if 'gt_labels_3d' in d:
                d['gt_labels_3d'] = DC(to_tensor(d['gt_labels_3d']))
            if 'gt_bboxes_3d' in d:
                b = d['gt_bboxes_3d']
                d['gt_bboxes_3d'] = DC(b, cpu_only=True) if isinstance(b, LiDARInstance3DBoxes) else DC(to_tensor(b))
        return d


@PIPELINES.register_module()
class Collect3D(Collect3D_GGA):
    """formating.py ``Collect3D``: the keys asked for + the meta container (the GGA variant adds nothing for these keys)."""
