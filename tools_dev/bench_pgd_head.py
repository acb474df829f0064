"""PGD head alone (forward + backward over the five FPN levels at bs 12) with one stream per level and with one stream."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, mono3d_heads
from gga_amd.cnn import to_channels_last
import bench

dev = torch.device('cuda:0')
cfg = Config.fromfile(bench.PGD_CONFIG if hasattr(bench, 'PGD_CONFIG') else os.path.join(os.path.dirname(bench.__file__), 'configs', 'gga', 'gga_pdg.py'))
torch.manual_seed(0)
model = to_channels_last(build_model(cfg.model).to(dev))
model.train()
head = model.bbox_head
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12       # 1: the maps are tiny, what is left is the host's launch time
sizes = head.featmap_sizes_of((B, 3, 384, 1248) if B > 1 else (1, 3, 64, 128))
feats = [torch.randn(B, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for h, w in sizes]


def step():
    outs = head(feats)
    loss = sum(t.float().sum() for lst in outs for t in lst if t is not None)
    loss.backward()


MODES = sys.argv[2].split(',') if len(sys.argv) > 2 else ['plain', 'streams', 'batch', 'plain', 'streams', 'batch']
for mode in MODES:
    mono3d_heads.LEVEL_STREAMS, mono3d_heads.LEVEL_BATCH = mode == 'streams', mode == 'batch'
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    print(f'{mode:8s} {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per head forward + backward', flush=True)
