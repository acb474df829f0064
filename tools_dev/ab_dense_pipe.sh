# A/B of the pipelined fragment sets of the dense 3x3 kernel (64-column forms): stand-alone launches and the PointPillars step,
# the shipped library against tools_dev/exp_libs/libgga_dc_nopipe.so (-DDC_PIPE_ON=0), alternating.
cat > /tmp/dc_time.py <<'P'
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from gga_amd import dense_conv
dense_conv.PLANES = 2
dev = 'cuda:0'
def t(B, C, Co, H, W):
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    trash = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    ts = []
    for i in range(14):
        trash.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = dense_conv._run(x, w, False)[0]; e1.record(); torch.cuda.synchronize()
        if i >= 4: ts.append(e0.elapsed_time(e1))
    print(f'   [{B},{C}->{Co},{H},{W}] {sum(ts) / len(ts) * 1e3:.0f} us (incl. absmax / pack)', float(y.abs().max()))
t(16, 64, 64, 248, 216); t(16, 384, 64, 248, 216); t(16, 128, 128, 124, 108)
P
pp() { python $1 bench.py --no-second-trunk --no-pgd --no-planes3 --no-fcaf3d --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   PP step', d['ms_per_step'])"; }
for i in 1 2; do
echo pipe; python /tmp/dc_time.py; pp ""
echo nopipe; python tools_dev/run_with_lib.py tools_dev/exp_libs/libgga_dc_nopipe.so /tmp/dc_time.py; pp "tools_dev/run_with_lib.py tools_dev/exp_libs/libgga_dc_nopipe.so"
done
