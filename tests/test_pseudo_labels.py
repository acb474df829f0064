"""Pseudo-label matching: golden vectors made by running the reference's
``pseudo_label_matching_kitti`` (tools/utils_pseudo_labels_gga.py:17-84) on
``synthetic.make_pseudo_case`` (tools_dev/make_golden.py::golden_pseudo_match). CPU: the oracle's
overlap/argmax against the golden overlaps. GPU: the product path (gga_image_box_match + host
bookkeeping) against every array of the golden re-labelled infos, bit for bit."""
import copy
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import REPO
from gga_amd import synthetic
from oracle import oracle as O

GOLD = np.load(os.path.join(REPO, 'tests', 'golden', 'pseudo_match.npz'))
CASES = (('f32', 31, 12, np.float32), ('f64', 32, 9, np.float64))


def _clean_gt_boxes(infos):
    out = []
    for info in infos:
        a = info['annos']
        n_obj = len([n for n in a['name'] if n != 'DontCare'])
        keep = [i for i in range(n_obj) if a['name'][i] in ('Pedestrian', 'Car', 'Cyclist')]
        out.append(a['bbox'][:n_obj][keep])
    return out


@pytest.mark.parametrize('cname,seed,nf,dtype', CASES)
def test_oracle_overlap_matches_reference(cname, seed, nf, dtype):
    infos, dts = synthetic.make_pseudo_case(seed, nf, dtype)
    gts = _clean_gt_boxes(infos)
    n_checked = 0
    for f in range(nf):
        want = GOLD[f'{cname}.{f}.overlap']
        got = O.image_box_overlap(dts[f]['bbox'], gts[f])
        assert got.dtype == want.dtype == dtype
        assert got.shape == want.shape
        assert np.array_equal(got, want), (cname, f)
        n_checked += got.size
    assert n_checked > 20


def test_oracle_first_maximum_rule():
    # two identical ground truths -> identical overlaps -> index of the first one
    gt = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [100, 100, 110, 110]], np.float64)
    dt = np.array([[1, 1, 9, 9], [200, 200, 210, 210]], np.float64)
    m = O.pseudo_label_match([dt], [gt])[0]
    assert m.tolist() == [0, 0]                      # no overlap at all -> argmax of zeros = 0
    ov = O.image_box_overlap(dt, gt)
    assert ov[0, 0] == ov[0, 1] == 64.0 / 100.0 and ov[0, 2] == 0.0


def test_drop_arrays_by_name():
    from gga_amd.pseudo_labels import drop_arrays_by_name
    got = drop_arrays_by_name(np.array(['Car', 'Van', 'Cyclist', 'DontCare', 'Pedestrian']))
    assert got.dtype == np.int64 and got.tolist() == [0, 2, 4]
    assert drop_arrays_by_name([]).shape == (0,)


@pytest.mark.gpu
@pytest.mark.parametrize('cname,seed,nf,dtype', CASES)
def test_pseudo_label_matching_kitti_vs_reference(cname, seed, nf, dtype, tmp_path):
    from gga_amd.pseudo_labels import pseudo_label_matching_kitti, image_box_match
    infos, dts = synthetic.make_pseudo_case(seed, nf, dtype)
    gts = _clean_gt_boxes(infos)
    match, ov = image_box_match([d['bbox'] for d in dts], gts, return_overlaps=True)
    for f in range(nf):
        want = GOLD[f'{cname}.{f}.overlap']
        assert ov[f].dtype == want.dtype and np.array_equal(ov[f], want), (cname, f)
        if want.shape[0]:
            assert np.array_equal(match[f], np.argmax(want, axis=-1))
    gi, di = copy.deepcopy(infos), copy.deepcopy(dts)
    out_file = str(tmp_path / 'pseudo' / 'infos.pkl')
    clean = pseudo_label_matching_kitti(gi, di, filename=out_file)
    assert clean[0] is gi[0]['annos']                # cleaned in place, like the reference
    with open(out_file, 'rb') as fh:
        dumped = pickle.load(fh)
    assert len(dumped) == nf
    for f in range(nf):
        assert set(dumped[f].keys()) == {'image', 'point_cloud', 'annos'}
        assert set(dumped[f]['annos'].keys()) == set(synthetic.PSEUDO_GT_KEYS)
        assert 'GGA_in_box_points' not in clean[f]
        for k in synthetic.PSEUDO_GT_KEYS:
            for tag, got in (('clean', clean[f][k]), ('new', dumped[f]['annos'][k])):
                want = GOLD[f'{cname}.{f}.{tag}.{k}']
                got = np.asarray(got)
                assert got.shape == want.shape, (cname, f, tag, k, got.shape, want.shape)
                assert got.dtype.kind == want.dtype.kind, (cname, f, tag, k)
                assert np.array_equal(got, want), (cname, f, tag, k)
    assert str(GOLD[f'{cname}.filename']) == './data/kitti_pesudo/kitti_infos_trainval_GGA_pseudo.pkl'


@pytest.mark.gpu
def test_image_box_match_large_batch_vs_oracle():
    # KITTI trainval scale: 7481 frames in one launch, against the C restatement
    rng = np.random.default_rng(5)
    dts, gts = [], []
    for f in range(7481):
        ng, nd = int(rng.integers(1, 15)), int(rng.integers(0, 20))
        g = rng.uniform(0, 1200, (ng, 2)); g = np.concatenate([g, g + rng.uniform(5, 300, (ng, 2))], 1)
        src = rng.integers(0, ng, nd)
        d = (g[src] + rng.uniform(-20, 20, (nd, 4))).astype(np.float32)
        dts.append(d); gts.append(g)
    from gga_amd.pseudo_labels import image_box_match
    got = image_box_match(dts, gts)
    want = O.pseudo_label_match(dts, gts)
    assert sum(len(w) for w in want) > 50000
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_image_box_match_errors():
    from gga_amd.pseudo_labels import image_box_match, pseudo_label_matching_kitti
    with pytest.raises(ValueError):                  # np.argmax over an empty axis in the reference
        image_box_match([np.zeros((2, 4), np.float32)], [np.zeros((0, 4))])
    assert image_box_match([], []) == []
    with pytest.raises(NotImplementedError):
        pseudo_label_matching_kitti([], [], metric=1)
