"""would skipping empty offsets pay in the halo form of the submanifold gather-GEMM on REAL scans? The bench's
synthetic frames scatter points uniformly in the volume; a LiDAR scan is a set of surfaces (ground, walls, object shells). For the
shipped config's sparse encoder this prints, per level and for both kinds of frame: rows, occupancy, populated offsets per ROW, and
per 32-row block of the halo form's Z-order tiles (the granularity at which the matrix instruction could skip an offset) the number
of offsets with at least one neighbour, the share of (block, offset) products that multiply nothing, and the share of blocks with
at least 9 empty offsets. Surface frames: a ground plane with 3 cm noise, four vertical walls, 12 box shells; same point count.
Usage (GPU box): python tools_dev/offset_stats_surface.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from gga_amd import Config, build_model, synthetic
from gga_amd.sparse import morton_order

DEV = 'cuda:0'
BS = 8


def surface_frame(seed, n=20000):
    rng = np.random.default_rng(seed)
    x0, y0, z0, x1, y1, z1 = synthetic.RANGE_SECOND
    parts = []
    ng = int(0.6 * n)                                           # ground: range-dependent density like a spinning LiDAR (1 / r)
    r = np.sqrt(rng.uniform(4.0, 70.0 ** 2, ng))
    a = rng.uniform(-0.75, 0.75, ng)
    parts.append(np.stack([r * np.cos(a), r * np.sin(a), -1.7 + rng.normal(0, 0.03, ng)], 1))
    nw = int(0.25 * n) // 4                                     # walls: vertical planes along / across the road
    for k in range(4):
        if k < 2:
            y = (-1) ** k * rng.uniform(8, 20)
            xs = rng.uniform(5, 65, nw)
            parts.append(np.stack([xs, np.full(nw, y) + rng.normal(0, 0.02, nw), rng.uniform(-1.7, 0.8, nw)], 1))
        else:
            x = rng.uniform(30, 65)
            ys = rng.uniform(-25, 25, nw)
            parts.append(np.stack([np.full(nw, x) + rng.normal(0, 0.02, nw), ys, rng.uniform(-1.7, 0.8, nw)], 1))
    no = n - sum(len(p) for p in parts)
    per = max(no // 12, 1)
    for k in range(12):                                         # object shells: points on the two faces towards the sensor
        c = np.array([rng.uniform(8, 55), rng.uniform(-15, 15), -1.7])
        l, w, h = 3.9 * rng.uniform(0.8, 1.2), 1.6 * rng.uniform(0.8, 1.2), 1.5
        u, v = rng.uniform(-0.5, 0.5, per), rng.uniform(0, 1, per)
        face = rng.random(per) < 0.5
        px = np.where(face, c[0] - l / 2, c[0] + u * l)
        py = np.where(face, c[1] + u * w, c[1] - np.sign(c[1] + 1e-6) * w / 2)
        parts.append(np.stack([px, py, c[2] + v * h], 1))
    pts = np.concatenate(parts)[:n]
    ok = (pts[:, 0] > x0) & (pts[:, 0] < x1) & (pts[:, 1] > y0) & (pts[:, 1] < y1) & (pts[:, 2] > z0) & (pts[:, 2] < z1)
    pts = pts[ok]
    return torch.from_numpy(np.concatenate([pts, rng.uniform(0, 1, (len(pts), 1))], 1).astype(np.float32))


cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
torch.manual_seed(0)
model = build_model(cfg.model).to(DEV).train()
enc = model.pts_middle_encoder
frames = {'uniform (bench)': [p.to(DEV) for p in synthetic.make_batch(BS, n_points=20000, pc_range=synthetic.RANGE_SECOND)['points']],
          'surfaces': [surface_frame(100 + i).to(DEV) for i in range(BS)]}
for kind, pts in frames.items():
    with torch.no_grad():
        v, n, c = model.voxelize(pts)
        plan = enc.build_indices(c, BS).index_plan
    levels = [plan.level0] + [entry[1] for entry in plan.indice_dict.values()]
    print(f'== {kind}: {sum(len(p) for p in pts)} points, {len(c)} voxels')
    for li, lvl in enumerate(levels):
        if lvl.n < 1024 or (3, 3, 3) not in lvl._subm:
            continue
        rb = lvl._subm[(3, 3, 3)]
        nbr = rb.nbr                                            # [27, n]
        has = nbr >= 0
        order = morton_order(lvl.coors).long()
        hz = has[:, order]
        nblk = lvl.n // 32
        blocks = hz[:, :nblk * 32].view(27, nblk, 32).any(2)    # [27, blocks]: offset populated in the block
        per_block = blocks.sum(0).float()
        cells = lvl.batch_size * lvl.shape[0] * lvl.shape[1] * lvl.shape[2]
        print(f'  level {li}: grid {lvl.shape} rows {lvl.n:7d} occupancy {lvl.n / cells:.4f}  offsets/row {float(has.sum()) / lvl.n:5.2f}  '
              f'offsets per 32-row Z-order block {float(per_block.mean()):5.2f} of 27  empty (block, offset) products {1 - float(per_block.mean()) / 27:.3f}  '
              f'blocks with >= 9 empty offsets {float((per_block <= 18).float().mean()):.3f}')
