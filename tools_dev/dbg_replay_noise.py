"""Latent-race hunt: twins of a short run of a LiDAR config, one quiet and the others with a second (high-priority) stream that keeps
small kernels running on the same CUs while the step executes. The step has no float atomics, so every twin must reproduce the
quiet run's losses bit for bit; a kernel with a missing barrier shows up as a twin that does not."""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench


def hunt(which='pp', steps=8, twins=40, batch=None, quiet=False):
    """-> number of noisy twins whose per-step losses differ from the quiet run's."""
    cfg, bs = (bench.PP_CONFIG, 16) if which == 'pp' else (bench.SECOND_CONFIG, 8)
    bs = batch or bs
    args = bench.parse_args(['--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-roofline'])
    torch.manual_seed(0)
    run = bench.run_workload(cfg, bs, 1, 0, args, 0, 1, torch.device('cuda:0'))
    runner, batches = run['runner'], run['batches']
    model = runner.raw_model
    state0 = copy.deepcopy(model.state_dict())
    opt0 = copy.deepcopy(runner.optimizer.state_dict())
    side = torch.cuda.Stream(priority=-1)
    noise = torch.randn(1 << 21, device='cuda:0')
    keys = torch.randint(0, 1 << 30, (1 << 19,), device='cuda:0')

    def disturb(n):
        with torch.cuda.stream(side):
            for _ in range(n):
                noise.mul_(1.0001).add_(1e-3)
                torch.sort(keys)
                torch.cumsum(noise, 0)

    def replay(noisy):
        model.load_state_dict(state0)
        runner.optimizer.load_state_dict(copy.deepcopy(opt0))
        runner.iter = 1
        runner._prepared.clear()
        torch.manual_seed(123)
        losses = []
        for i in range(steps):
            if noisy:
                disturb(noisy)
            out = runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
            losses.append(out['loss'].detach())
        torch.cuda.synchronize()
        return torch.stack(losses).double().cpu()

    a = replay(0)
    bad = 0
    for rep in range(twins):
        b = replay(20 + 10 * (rep % 4))
        d = (a != b).nonzero()
        if len(d):
            bad += 1
            i = int(d[0])
            if not quiet:
                print(f'twin {rep}: differs from step {i}: {float(a[i]):.9g} vs {float(b[i]):.9g}')
    return bad


if __name__ == '__main__':
    which = os.environ.get('GGA_REPLAY_CFG', 'pp')
    twins = int(os.environ.get('GGA_REPLAY_TWINS', '40'))
    bad = hunt(which, int(os.environ.get('GGA_REPLAY_STEPS', '8')), twins)
    print(f'{which}: {bad} of {twins} noisy twins differ from the quiet run')
