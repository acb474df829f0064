"""Worker of tests/test_model_gpu.py::test_ddp_over_rccl_one_rank_is_bit_identical: ONE rank, backend ``nccl`` (= RCCL), on the one
device of the box. A world of one still takes DistributedDataParallel's whole path: the reducer's bucket hooks fire inside the
backward, every bucket is all-reduced on RCCL's own stream, and the backward's end waits for those streams before the optimizer
reads the gradients - the stream hand-overs gloo (whose all-reduce goes through the host and synchronises) cannot exercise. The
step's own side streams are on: the head branches on two streams (GGA_HEAD_STREAMS=2), the prefetched front of the next batch on
its high-priority stream, and a background thread keeping small kernels on a third one.

No kernel of the step sums with float atomics and the all-reduce of one rank is the identity (then a division by 1), so after
``STEPS`` optimizer steps the DDP-wrapped run must hold the SAME BITS as the plain Runner - parameters, BatchNorm buffers and
losses; a missing dependency between a side stream and the bucket's all-reduce shows up as a difference (or as a twin of the
wrapped run that differs from itself). Prints one line per config; exit code 0 only when everything is identical."""
import copy
import os
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('GGA_HEAD_STREAMS', '2')
import torch
import torch.distributed as dist

from gga_amd import Config, build_model, synthetic
from gga_amd.cnn import to_channels_last
from gga_amd.train import Runner

STEPS = 5
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29631'), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1)
assert dist.get_backend() == 'nccl'

stop = threading.Event()


def noise():
    side = torch.cuda.Stream(priority=-1)
    buf = torch.randn(1 << 21, device=dev)
    keys = torch.randint(0, 1 << 30, (1 << 19,), device=dev)
    while not stop.is_set():
        with torch.cuda.stream(side):
            for _ in range(8):
                buf.mul_(1.0001).add_(1e-3)
                torch.sort(keys)
                torch.cumsum(buf, 0)
        side.synchronize()
        time.sleep(0.0005)


thread = threading.Thread(target=noise, daemon=True)
thread.start()


def run(cfg, model, batches, distributed):
    runner = Runner(model, cfg, max_iters=100, distributed=distributed, device=dev)
    assert runner.planes == 2 or os.environ.get('GGA_DENSE_PLANES')
    runner.inputs_ready(*batches)
    losses = []
    for i in range(STEPS):
        out = runner.step(batches[i % len(batches)], next_data=batches[(i + 1) % len(batches)])
        losses.append(out['loss'].detach().clone())
    torch.cuda.synchronize()
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return torch.stack(losses).cpu(), state, runner


ok = True
for name, path, rng, B in (('pp', 'gga_kitti_pointpillars_config.py', synthetic.RANGE_PP, 4), ('second', 'gga_kitti_config.py', synthetic.RANGE_SECOND, 2)):
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', path))
    cfg.model.pts_middle_encoder['channels_last'] = True
    torch.manual_seed(5)
    base = to_channels_last(build_model(cfg.model).to(dev)).train()
    with torch.no_grad():
        for th in base.pts_bbox_head.task_heads:
            for n in ('reg', 'height', 'dim', 'rot'):
                getattr(th, n)[-1].weight.mul_(0.05)
    batches = []
    for i in range(2):
        b = synthetic.make_batch(B, start=20 + B * i, n_points=20000, pc_range=rng)
        b['points'] = [p.to(dev) for p in b['points']]
        batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
    # the SRL coefficients are drawn from the CPU generator once per step (centerpoint_head_gga.py:514-525): the runs start
    # from the same generator state
    runs = []
    for distributed in (False, True, True):
        torch.manual_seed(11)
        runs.append(run(cfg, copy.deepcopy(base), batches, distributed))
    (l0, s0, _), (l1, s1, r1), (l2, s2, _) = runs
    from torch.nn.parallel import DistributedDataParallel
    assert isinstance(r1.model, DistributedDataParallel) and r1.model.module is r1.raw_model
    diff_params = [k for k in s0 if not torch.equal(s0[k], s1[k])]
    diff_twin = [k for k in s1 if not torch.equal(s1[k], s2[k])]
    same_loss = torch.equal(l0, l1) and torch.equal(l1, l2)
    good = not diff_params and not diff_twin and same_loss and bool(torch.isfinite(l0).all())
    ok = ok and good
    print(f'NCCL1 {name}: steps {STEPS} tensors {len(s0)} losses {[round(float(x), 6) for x in l0]} identical_losses {same_loss} '
          f'ddp_vs_plain_differing {diff_params[:4]} ({len(diff_params)}) ddp_twin_differing {diff_twin[:4]} ({len(diff_twin)}) '
          f'head_streams {os.environ["GGA_HEAD_STREAMS"]} backend {dist.get_backend()} ok {good}', flush=True)
    del runs, base
    torch.cuda.empty_cache()
stop.set()
thread.join(timeout=10)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
