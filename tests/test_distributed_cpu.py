"""N>1 path on CPU: two gloo ranks through the product's Runner / build_ddp
(gga_amd/train.py) — frames are sharded by rank, the only exchange is the gradient all-reduce,
and a DDP step equals the single-process step on the averaged loss."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from gga_amd import Config, synthetic
from gga_amd.detectors import MVXTwoStageDetector_GGA
from gga_amd.train import Runner

CFG = dict(optimizer=dict(type='AdamW', lr=1e-2, betas=(0.95, 0.99), weight_decay=0.01),
           optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)),
           lr_config=dict(policy='cyclic', target_ratio=(10, 1e-4), cyclic_times=1, step_ratio_up=0.4),
           momentum_config=dict(policy='cyclic', target_ratio=(0.85 / 0.95, 1), cyclic_times=1, step_ratio_up=0.4))


class TinyDet(nn.Module):
    """CPU stand-in with the detector's train-step surface (the real stages need the GPU)."""
    _parse_losses = MVXTwoStageDetector_GGA._parse_losses
    train_step = MVXTwoStageDetector_GGA.train_step

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.net = nn.Sequential(nn.Linear(4, 16), nn.ReLU(), nn.Linear(16, 1))

    def forward(self, return_loss=True, points=None, img_metas=None, **kw):
        out = torch.stack([self.net(p).mean() for p in points])
        return {'task0.loss_heatmap': out.pow(2).mean(), 'task0.distancemin': out.abs().mean().detach()}


def _data(rank, frames=2):
    b = synthetic.make_batch(frames, start=0, rank=rank, n_points=300, n_obj_range=(2, 3), n_ibp_range=(5, 10))
    return dict(points=b['points'], img_metas=b['img_metas'])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = TinyDet()
    runner = Runner(model, Config(CFG), max_iters=10, distributed=True, device=torch.device('cpu'))
    out = None
    for _ in range(3):
        out = runner.step(_data(rank))
    q.put((rank, [p.detach().numpy().copy() for p in model.parameters()], float(out['loss'].detach())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_step_matches_single_process():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # ranks saw different frames (sharding by rank seed) ...
    assert not torch.equal(_data(0)['points'][0], _data(1)['points'][0])
    assert res[0][2] != res[1][2]
    # ... but hold identical parameters after the all-reduced steps
    for a, b in zip(res[0][1], res[1][1]):
        assert (a == b).all()
    # and equal the single-process run on the mean of the two ranks' losses
    model = TinyDet()
    runner = Runner(model, Config(CFG), max_iters=10)
    for _ in range(3):
        for g in runner.optimizer.param_groups:
            g['lr'] = runner.lr_sched(runner.iter)
            g['betas'] = (runner.mom_sched(runner.iter), g['betas'][1])
        losses = [model._parse_losses(model(**_data(r)))[0] for r in range(2)]
        runner.optimizer.zero_grad()
        (sum(losses) / 2).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 35, 2)
        runner.optimizer.step()
        runner.iter += 1
    for a, b in zip(res[0][1], model.parameters()):
        torch.testing.assert_close(torch.from_numpy(a), b.detach(), rtol=1e-5, atol=1e-7)


# ---------------------------------------------------------------------------------------------------- multi-GPU test run
class _Frames(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from gga_amd.pipelines import DataContainer as DC
        return dict(points=[DC(torch.full((3 + i, 4), float(i)))], img_metas=[DC(dict(sample_idx=i), cpu_only=True)])


class _EchoDet(nn.Module):
    """Test-mode stand-in: one result dict per frame, naming the frame it saw."""

    def forward(self, return_loss=True, rescale=False, points=None, img_metas=None, **kw):
        assert not return_loss and rescale
        return [dict(sample_idx=m['sample_idx'], n_points=int(p.shape[0]), mean=float(p.mean()))
                for p, m in zip(points[0], img_metas[0])]


def _test_worker(rank, world, port, q, tmpdir, gpu_collect):
    from gga_amd.apis import multi_gpu_test
    from gga_amd.loader import build_dataloader
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    loader = build_dataloader(_Frames(7), samples_per_gpu=2, workers_per_gpu=0, dist=True, shuffle=False)
    out = multi_gpu_test(_EchoDet(), loader, tmpdir=tmpdir, gpu_collect=gpu_collect, device=torch.device('cpu'))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('gpu_collect', [False, True])
def test_two_rank_test_run_returns_the_dataset_in_order(gpu_collect, tmp_path):
    """The pseudo-label run over several ranks (reference: tools/generate_pseudo_labels_gga.py:242 ``multi_gpu_test``,
    tools/dist_pseudo.sh): 7 frames over 2 ranks in batches of 2 - strided shards, one padded repeat - come back on rank 0 as
    7 results in dataset order, through files or through the process group; the other rank gets None."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_test_worker, args=(r, 2, port, q, None if gpu_collect else str(tmp_path / 'collect'), gpu_collect))
             for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None
    assert [r['sample_idx'] for r in res[0]] == list(range(7))
    assert [r['n_points'] for r in res[0]] == [3 + i for i in range(7)] and [r['mean'] for r in res[0]] == [float(i) for i in range(7)]
    assert not (tmp_path / 'collect').exists()


def _seed_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import numpy as np
    from gga_amd.train import init_random_seed
    np.random.seed(100 + rank)                      # the ranks' own generators differ: what comes back must not
    q.put((rank, init_random_seed(None, device='cpu'), init_random_seed(5, device='cpu')))
    dist.barrier()
    dist.destroy_process_group()


def test_init_random_seed_is_common_to_the_ranks():
    """``init_random_seed`` (mmdet3d/apis/train.py:27-55): without a seed rank 0 draws one and broadcasts it - the one collective
    of the start-up -, with a seed every rank returns it."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_seed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and got[0][2] == got[1][2] == 5
