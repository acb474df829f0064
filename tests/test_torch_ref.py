"""The torch fp32 restatement (oracle/torch_ref.py) against the imported reference's golden
losses AND autograd gradients. CPU only."""
import numpy as np
import pytest
import torch

from conftest import FMAP, TRAIN_CFG, load_head_case
from gga_amd import synthetic
from oracle import oracle as O
from oracle import torch_ref as R


@pytest.mark.parametrize('c', ['second', 'pp'])
def test_head_loss_and_grads(golden, c):
    d = golden('head')
    case = load_head_case(d, c)
    torch.manual_seed(1234)
    srl = O.draw_srl(case['B'])
    tg = O.get_targets(case['labels'], case['boxes_img'], case['lidar2img'], case['pseudo'],
                       case['bdry'], case['ibp'], case['meta_l2i'], TRAIN_CFG[c], srl)
    H, W = FMAP[c]
    preds = [{k: v.requires_grad_(True) for k, v in p.items()}
             for p in synthetic.make_head_preds(case['B'], H, W, seed=int(d[f'{c}.pred_seed']))]
    losses = R.head_loss(preds, tg, TRAIN_CFG[c])
    assert len(losses) == 18
    for k, v in losses.items():
        assert float(v) == pytest.approx(float(d[f'{c}.loss.{k}']), rel=1e-5, abs=1e-4), k
    tot_a = sum(v for k, v in losses.items() if 'loss' in k)
    tot_b = sum(losses.values())
    assert float(tot_a) == pytest.approx(float(d[f'{c}.total_A']), rel=1e-5)
    leaves = [preds[t][k] for t in range(3) for k in ('reg', 'height', 'dim', 'rot', 'heatmap')]
    ga = torch.autograd.grad(tot_a, leaves, retain_graph=True, allow_unused=True)
    gb = torch.autograd.grad(tot_b, leaves, allow_unused=True)
    i = 0
    for t in range(3):
        for k in ('reg', 'height', 'dim', 'rot', 'heatmap'):
            if k == 'heatmap':
                sel = d[f'{c}.gradA.{t}.heatmap.flatidx']
                np.testing.assert_allclose(ga[i].numpy().reshape(-1)[sel], d[f'{c}.gradA.{t}.heatmap.val'],
                                           rtol=1e-4, atol=1e-7)
            else:
                for tag, g in (('A', ga[i]), ('B', gb[i])):
                    idx = d[f'{c}.grad{tag}.{t}.{k}.idx']
                    got = g.numpy()[tuple(idx.T)] if len(idx) else np.zeros(0, np.float32)
                    np.testing.assert_allclose(got, d[f'{c}.grad{tag}.{t}.{k}.val'], rtol=1e-3, atol=1e-5)
            i += 1


def test_scatter_matches_golden(golden):
    d = golden('scatter')
    B, C, ny, nx = (int(x) for x in d['pp.shape'])
    out = R.scatter(torch.from_numpy(d['pp.feats']), torch.from_numpy(d['pp.coors']), B, ny, nx)
    assert np.array_equal(out.numpy(), d['pp.canvas'])
