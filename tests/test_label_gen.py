"""Offline GGA label generation primitives against vectors produced by the
reference's tools/data_converter/utils_gga.py (tools_dev/make_golden.py::golden_label_gen):
region_grow for 7 thresholds x 3 ratio settings x 3 point sets, points_in_frustm_indices, and
calculate_ground with equal numpy seeds - all masks bit for bit. CPU: the oracle; GPU: the product."""
import os
import sys

import numpy as np
import pytest

from conftest import REPO
from gga_amd import synthetic
from oracle import oracle as O

sys.path.insert(0, os.path.join(REPO, 'tools_dev'))
import make_golden as MG  # noqa: E402

GOLD = np.load(os.path.join(REPO, 'tests', 'golden', 'label_gen.npz'))


def _bits(key, n):
    return np.unpackbits(GOLD[key])[:n].astype(bool)


def _check_region_grow(fn_multi):
    checked = 0
    for seed in MG.LABEL_GEN_SEEDS:
        pc, ms, mo = synthetic.make_region_grow_case(seed)
        for ratio in MG.LABEL_GEN_RATIOS:
            got = fn_multi(pc, ms, mo, [(j + 1) * 0.1 for j in range(7)], ratio)
            for j in range(7):
                want = _bits(f'rg.{seed}.{ratio}.{j}', len(pc))
                assert got[j].dtype == np.float64 and np.array_equal(got[j].astype(bool), want), (seed, ratio, j)
                checked += int(want.sum())
    assert checked > 3000


def test_oracle_region_grow_matches_reference():
    _check_region_grow(lambda pc, ms, mo, ths, ratio: [O.region_grow(pc, ms, mo, t, ratio) for t in ths])


def _check_frustum(fn):
    c = synthetic.KITTI_CALIB
    for seed in MG.LABEL_GEN_SEEDS[:2]:
        pts, boxes = synthetic.make_frustum_case(seed)
        for b, box in enumerate(boxes):
            got = fn(pts, c['R0_rect'], c['Tr_velo_to_cam'], c['P2'], box)
            assert got.shape == (len(pts), 1) and got.dtype == bool
            assert np.array_equal(got.squeeze(), _bits(f'fr.{seed}.{b}', len(pts))), (seed, b)


def test_oracle_frustum_matches_reference():
    _check_frustum(O.points_in_frustm_indices)


def _check_ground(fn):
    for seed in MG.LABEL_GEN_SEEDS[:2]:
        cloud = synthetic.make_ground_case(seed)
        np.random.seed(seed)
        mask, tri = fn(cloud, 0.2)
        assert np.array_equal(mask.astype(bool), _bits(f'gr.{seed}.mask', len(cloud))) and np.array_equal(tri, GOLD[f'gr.{seed}.triple'])
        np.random.seed(seed)
        mask, tri = fn(cloud, 0.15, back_cut=True, back_cut_z=10.0)
        n_cut = int((cloud[:, 2] > 10.0).sum())
        assert len(mask) == n_cut
        assert np.array_equal(mask.astype(bool), _bits(f'gr.{seed}.mask_cut', n_cut)) and np.array_equal(tri, GOLD[f'gr.{seed}.triple_cut'])


def test_oracle_ground_matches_reference():
    _check_ground(O.calculate_ground)


@pytest.mark.gpu
def test_region_grow_matches_reference():
    from gga_amd import label_gen as LG
    _check_region_grow(LG.region_grow_multi)
    pc, ms, mo = synthetic.make_region_grow_case(61)
    assert np.array_equal(LG.region_grow(pc, ms, mo, 0.3, 0.85), LG.region_grow_multi(pc, ms, mo, [0.3], 0.85)[0])
    assert LG.region_grow(pc, np.zeros(len(pc)), np.zeros(len(pc)), 0.3).sum() == 0       # empty search set


@pytest.mark.gpu
def test_region_grow_kitti_scale_vs_oracle():
    # one object at KITTI scale: ~20 k search points, a 600-point object and a neighbour 0.5 m away
    rng = np.random.default_rng(3)
    n = 20000
    pc = np.stack([rng.uniform(-20, 20, n), rng.uniform(-1, 2, n), rng.uniform(3, 60, n), np.ones(n)], 1)
    pc[:600, :3] = np.array([2.0, 1.0, 15.0]) + rng.normal(0, [0.6, 0.4, 0.9], (600, 3))
    pc[600:900, :3] = np.array([4.6, 1.0, 15.5]) + rng.normal(0, [0.4, 0.4, 0.5], (300, 3))
    ms = (rng.random(n) < 0.9).astype(np.float64)
    mo = ms * ((np.abs(pc[:, 0] - 2.0) < 1.6) & (np.abs(pc[:, 2] - 15.0) < 2.4))
    from gga_amd import label_gen as LG
    ths = [(j + 1) * 0.1 for j in range(7)]
    got = LG.region_grow_multi(pc, ms, mo, ths, 0.85)
    for j, t in enumerate(ths):
        assert np.array_equal(got[j], O.region_grow(pc, ms, mo, t, 0.85)), j
    assert got.sum(1).max() > 300            # 0.3 m isolates the object; from 0.4 m on it merges with the neighbour and fails the ratio
    best = got[int(np.argmax(got.sum(1)))]
    grown = LG.region_grow(pc, np.ones(n), best, 0.3, ratio=None)            # the un-truncated cluster (converter :401-405)
    assert np.array_equal(grown, O.region_grow(pc, np.ones(n), best, 0.3, None)) and grown.sum() >= best.sum()


@pytest.mark.gpu
def test_frustum_matches_reference():
    from gga_amd import label_gen as LG
    _check_frustum(LG.points_in_frustm_indices)


@pytest.mark.gpu
def test_ground_matches_reference():
    from gga_amd import label_gen as LG
    _check_ground(LG.calculate_ground)


def test_fit_pseudo_box_properties():
    # no callable in the reference to pin this against (inline code of _calculate_rga): geometric properties
    from gga_amd import label_gen as LG
    rng = np.random.default_rng(0)
    for yaw in (0.0, 0.3, 1.0, 1.4):
        l, w = 4.2, 1.7
        local = np.stack([rng.uniform(-l / 2, l / 2, 400), rng.uniform(-w / 2, w / 2, 400)], 1)
        local[:4] = [[-l / 2, -w / 2], [l / 2, -w / 2], [l / 2, w / 2], [-l / 2, w / 2]]
        c, s = np.cos(yaw), np.sin(yaw)
        xy = local @ np.array([[c, s], [-s, c]]) + np.array([12.0, -3.0])
        clt = np.concatenate([xy, rng.uniform(-1.6, -0.2, (400, 1)), np.ones((400, 1))], 1)
        box, centre, rot = LG.fit_pseudo_box(clt, -1.7)
        assert box.shape == (1, 7) and centre.shape == (1, 2)
        assert box[0, 3] >= box[0, 4]                                         # longer side first
        assert abs(box[0, 3] - l) < 0.25 and abs(box[0, 4] - w) < 0.25
        assert np.allclose(box[0, :2], [12.0, -3.0], atol=0.15)
        assert abs(box[0, 5] - (clt[:, 2].max() + 1.7)) < 1e-12 and abs(box[0, 2] - (clt[:, 2].max() - 1.7) / 2) < 1e-12
        # the heading lies on the 2.5 degree grid next to the true one (mod 90 degrees: longer side first)
        d = (rot - yaw) % (np.pi / 2)
        assert min(d, np.pi / 2 - d) < np.pi / 72 + 1e-6, (yaw, rot)


# ----------------------------------------------------------------------------- whole-frame generator
RGA = np.load(os.path.join(REPO, 'tests', 'golden', 'rga.npz'))


def test_box2d_labels_and_pseudo_box_fit_match_reference_run():
    # host-side pieces of calculate_rga against the reference's _calculate_rga run: the 2D-box labels
    # (the reference used this repo's stand-ins for the absent nuscenes / shapely pieces there, so
    # this is a consistency check) and the pseudo 3D box fit on the reference's own in-box points
    # (independent: that code is the reference's)
    from gga_amd import label_gen as LG
    for seed in MG.RGA_SEEDS:
        pts, calib, annos, shape = synthetic.make_rga_scene(seed)
        n_obj = len([n for n in annos['name'] if n != 'DontCare'])
        boxes, depth, m2d, mb, bdry = LG.box2d_labels(annos, calib['P2'], shape)
        # corners are rotated in float32 by torch on the host (array_converter semantics): sin / cos / einsum
        # of another CPU may differ in the last float32 bit, so these two labels carry a float32 tolerance
        assert np.allclose(boxes, RGA[f'{seed}.GGA_boxes_img'][:n_obj], rtol=2e-6, atol=1e-4)
        assert np.array_equal(depth, RGA[f'{seed}.GGA_mask_depth'][:n_obj]) and np.array_equal(m2d, RGA[f'{seed}.GGA_mask2d'][:n_obj])
        assert np.array_equal(mb, RGA[f'{seed}.GGA_mask_boundary'][:n_obj]) and np.array_equal(bdry, RGA[f'{seed}.GGA_bdry_masks'][:n_obj])
        # ground height as the reference computed it is implied by the golden boxes: z centre + dz/2 = cluster top
        lens, cat = RGA[f'{seed}.in_box_len'], RGA[f'{seed}.in_box_cat']
        off = 0
        for i, n in enumerate(lens):
            if n == 0:
                continue
            clt = cat[off:off + n]; off += n
            want = RGA[f'{seed}.GGA_init_pseudo_label'][i]
            ground = clt[:, 2].max() - want[5]
            box, _, _ = LG.fit_pseudo_box(clt, ground)
            assert np.allclose(box[0, [0, 1, 3, 4, 6]], want[[0, 1, 3, 4, 6]], rtol=2e-6, atol=1e-5), (seed, i)


@pytest.mark.gpu
def test_calculate_rga_matches_reference_run():
    from gga_amd import label_gen as LG
    for seed in MG.RGA_SEEDS:
        pts, calib, annos, shape = synthetic.make_rga_scene(seed)
        np.random.seed(seed)
        out = LG.calculate_rga(pts, calib, annos, shape)
        assert out is annos
        for k in MG.RGA_KEYS:
            got, want = np.asarray(out[k]), RGA[f'{seed}.{k}']
            assert got.shape == want.shape and got.dtype == want.dtype, (seed, k, got.dtype, want.dtype)
            if k in ('GGA_boxes_img', 'GGA_init_pseudo_label'):      # host float32 rotations: see the test above
                assert np.allclose(got, want, rtol=2e-6, atol=1e-4), (seed, k)
            else:
                assert np.array_equal(got, want), (seed, k)
        lens = np.array([len(p) for p in out['GGA_in_box_points']], np.int64)
        assert np.array_equal(lens, RGA[f'{seed}.in_box_len'])
        cat = [np.asarray(p) for p in out['GGA_in_box_points'] if len(p)]
        assert np.array_equal(np.concatenate(cat, 0), RGA[f'{seed}.in_box_cat'])
        assert not out['GGA_mask_boundary'][1] and out['GGA_mask_valid'][1]      # the border-cut object took the ratio=None branch
