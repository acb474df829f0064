# Ablation builds of the dense 3x3 kernel (tools_dev/exp_libs/libgga_abl_*.so, built from dense_conv.hip with -DABL_*=1),
# each timed on the 128-column shapes of both configs next to the shipped library.
cat > /tmp/abl_time.py <<'P'
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from gga_amd import dense_conv
dense_conv.PLANES = 2
dev = 'cuda:0'
out = []
def t(B, C, Co, H, W):
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    xa, wa = dense_conv._amax_bits(x), dense_conv._amax_bits(w)
    trash = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    ts = []
    for i in range(12):
        trash.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = dense_conv._run(x, w, False, x_amax=xa, w_amax=wa)[0]; e1.record(); torch.cuda.synchronize()
        if i >= 4: ts.append(e0.elapsed_time(e1))
    ts.sort()
    out.append(f'{ts[len(ts) // 2] * 1e3:6.0f}')
for shp in ((16, 128, 128, 124, 108), (8, 128, 128, 200, 176), (16, 256, 256, 62, 54), (16, 64, 64, 248, 216)):
    t(*shp)
print(' '.join(out))
P
echo "variant: us at 16x128->128x124x108 (16-row)  8x128->128x200x176 (8-row)  16x256->256x62x54 (8-row)  16x64->64x248x216 (64-col)"
for rep in 1 2; do
printf "%-12s" shipped; python /tmp/abl_time.py
for v in ${VARIANTS:-nostore nohalo now nobar noread mem nomemnobar all}; do
printf "%-12s" $v; python tools_dev/run_with_lib.py tools_dev/exp_libs/libgga_abl_$v.so /tmp/abl_time.py
done
done
