"""Twins of a 25-step run without any instrumentation (only the step losses are kept, on the device): how often do two runs of the
same inputs from the same state differ, and from which step on?"""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
STEPS = int(os.environ.get('GGA_REPLAY_STEPS', '25'))
TWINS = int(os.environ.get('GGA_REPLAY_TWINS', '16'))
args = bench.parse_args(['--batch', '8', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-roofline'])
torch.manual_seed(0)
run = bench.run_workload(bench.SECOND_CONFIG, 8, 1, 0, args, 0, 1, torch.device('cuda:0'))
runner, batches = run['runner'], run['batches']
model = runner.raw_model
state0 = copy.deepcopy(model.state_dict())
opt0 = copy.deepcopy(runner.optimizer.state_dict())


def replay():
    model.load_state_dict(state0)
    runner.optimizer.load_state_dict(copy.deepcopy(opt0))
    runner.iter = 1
    runner._prepared.clear()
    torch.manual_seed(123)
    losses = []
    for i in range(STEPS):
        out = runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
        losses.append(out['loss'].detach())
    torch.cuda.synchronize()
    return torch.stack(losses).double().cpu()


a = replay()
bad = 0
for rep in range(TWINS):
    b = replay()
    d = (a != b).nonzero()
    if len(d):
        bad += 1
        i = int(d[0])
        print(f'twin {rep}: differs from step {i}: {float(a[i]):.9g} vs {float(b[i]):.9g}; final {float(a[-1]):.6g} vs {float(b[-1]):.6g}')
print(f'{bad} of {TWINS} twins differ; reference final {float(a[-1]):.6g}')
