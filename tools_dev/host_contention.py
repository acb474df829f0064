"""Launch-side host cost of a train step with 1 and with 8 ranks on one host (SURVEY.md 5 names the Python launch side as the
scaling risk at 8 ranks per node). Every rank steps the PointPillars GGA model at batch 1 (device time per step far below
host time, so the step is launch-bound and its wall time IS the host time) under DistributedDataParallel over gloo, all
ranks sharing the one GPU of the box; reported per rank: wall ms / step and CPU ms / step (time.process_time).
    python tools_dev/host_contention.py            # parent: runs 1 rank, then 8 ranks, prints / writes the summary"""
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def worker():
    import torch
    import torch.distributed as dist
    from gga_amd import Config, build_model, synthetic
    from gga_amd.cnn import to_channels_last
    from gga_amd.train import Runner, init_dist, setup_multi_processes
    rank, world, _ = init_dist()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_pointpillars_config.py'))
    setup_multi_processes(cfg)
    cfg.model.pts_middle_encoder['channels_last'] = True
    torch.manual_seed(0)
    model = to_channels_last(build_model(cfg.model).to(dev)).train()
    with torch.no_grad():
        for th in model.pts_bbox_head.task_heads:
            for name in ('reg', 'height', 'dim', 'rot'):
                getattr(th, name)[-1].weight.mul_(0.05)
    runner = Runner(model, cfg, max_iters=1000, distributed=world > 1, device=dev)
    b = synthetic.make_batch(1, rank=rank, n_points=20000, pc_range=synthetic.RANGE_PP)
    b['points'] = [p.to(dev) for p in b['points']]
    data = {k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)}
    for _ in range(10):
        runner.step(data)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    runner.freeze_gc()
    n = 40
    w0, c0 = time.perf_counter(), time.process_time()
    for _ in range(n):
        runner.step(data)
    w1, c1 = time.perf_counter(), time.process_time()          # launch side only: nothing waits for the device here
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    print('HOST ' + json.dumps(dict(rank=rank, world=world, launch_wall_ms=(w1 - w0) / n * 1e3, launch_cpu_ms=(c1 - c0) / n * 1e3,
                                    wall_ms_with_device=(w2 - w0) / n * 1e3)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def launch(n):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GGA_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', GGA_HOST_WORKER='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    rows = [json.loads(l[5:]) for l in out.stdout.splitlines() if l.startswith('HOST ')]
    assert len(rows) == n, (out.stdout[-2000:], out.stderr[-2000:])
    return rows


if __name__ == '__main__':
    if os.environ.get('GGA_HOST_WORKER') == '1':
        worker()
    else:
        res = {}
        summ = lambda rows: dict(launch_wall_ms_mean=sum(r['launch_wall_ms'] for r in rows) / len(rows),
                                 launch_wall_ms_max=max(r['launch_wall_ms'] for r in rows),
                                 launch_cpu_ms_mean=sum(r['launch_cpu_ms'] for r in rows) / len(rows),
                                 wall_ms_with_device_max=max(r['wall_ms_with_device'] for r in rows))
        # (a) N independent processes (no process group): what the launch side of one rank costs next to N - 1 others. They share
        #     the one GPU, so with 8 of them the device is the bottleneck and wall time measures its queue; CPU ms is the figure
        for n in (1, 8):
            env = dict(os.environ, GGA_HOST_WORKER='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(env, GGA_HOST_SEED=str(i)), stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL, text=True) for i in range(n)]
            rows = []
            for p_ in procs:
                out, _ = p_.communicate(timeout=1200)
                rows += [json.loads(l[5:]) for l in out.splitlines() if l.startswith('HOST ')]
            assert len(rows) == n
            res[f'independent_processes_{n}'] = summ(rows)
        # (b) the same under DistributedDataParallel over gloo (2 ranks): gloo's all-reduce of the 5.6 M gradients goes through host
        #     memory and host threads - a functional stand-in for RCCL, its cost says nothing about RCCL's
        res['ddp_gloo_ranks_2'] = summ(launch(2))
        res['how'] = __doc__.split('\n    python')[0].replace('\n', ' ')
        res['host_threads'] = os.cpu_count()
        try:
            res['load_average'] = os.getloadavg()
        except OSError:
            pass
        os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
        json.dump(res, open(os.path.join(REPO, 'gpurun_out', 'host_contention.json'), 'w'), indent=1)
        print(json.dumps(res, indent=1))
