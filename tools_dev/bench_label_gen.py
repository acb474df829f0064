#!/usr/bin/env python
"""Label-generation primitives at KITTI frame scale: device vs the C oracle (1 host thread)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from gga_amd import label_gen as LG, synthetic
from oracle import oracle as O
rng = np.random.default_rng(3)
n = 20000
pc = np.stack([rng.uniform(-20, 20, n), rng.uniform(-1, 2, n), rng.uniform(3, 60, n), np.ones(n)], 1)
pc[:600, :3] = np.array([2.0, 1.0, 15.0]) + rng.normal(0, [0.6, 0.4, 0.9], (600, 3))
pc[600:900, :3] = np.array([4.6, 1.0, 15.5]) + rng.normal(0, [0.4, 0.4, 0.5], (300, 3))
ms = (rng.random(n) < 0.9).astype(np.float64)
mo = ms * ((np.abs(pc[:, 0] - 2.0) < 1.6) & (np.abs(pc[:, 2] - 15.0) < 2.4))
ths = [(j + 1) * 0.1 for j in range(7)]
LG.region_grow_multi(pc, ms, mo, ths, 0.85)
t0 = time.perf_counter(); got = LG.region_grow_multi(pc, ms, mo, ths, 0.85); tg = time.perf_counter() - t0
t0 = time.perf_counter(); [O.region_grow(pc, ms, mo, t, 0.85) for t in ths]; tc = time.perf_counter() - t0
print(f'region_grow, 7 thresholds, {int(ms.sum())} search / {int(mo.sum())} origin points: device {tg * 1e3:.1f} ms (incl. H2D/D2H), oracle {tc * 1e3:.0f} ms')
c = synthetic.KITTI_CALIB
pts = np.stack([rng.uniform(0, 70, 120000), rng.uniform(-40, 40, 120000), rng.uniform(-3, 1, 120000), np.ones(120000)], 1)
box = np.array([300.0, 120.0, 520.0, 260.0])
LG.points_in_frustm_indices(pts, c['R0_rect'], c['Tr_velo_to_cam'], c['P2'], box)
t0 = time.perf_counter(); LG.points_in_frustm_indices(pts, c['R0_rect'], c['Tr_velo_to_cam'], c['P2'], box); tg = time.perf_counter() - t0
t0 = time.perf_counter(); O.points_in_frustm_indices(pts, c['R0_rect'], c['Tr_velo_to_cam'], c['P2'], box); tc = time.perf_counter() - t0
print(f'points_in_frustm_indices, 120 k points: device {tg * 1e3:.2f} ms (incl. H2D/D2H), oracle {tc * 1e3:.1f} ms')
cloud = synthetic.make_ground_case(5, n=120000)
np.random.seed(0); LG.calculate_ground(cloud, 0.2)
np.random.seed(0); t0 = time.perf_counter(); LG.calculate_ground(cloud, 0.2); tg = time.perf_counter() - t0
np.random.seed(0); t0 = time.perf_counter(); O.calculate_ground(cloud, 0.2); tc = time.perf_counter() - t0
print(f'calculate_ground (5 x 100 RANSAC planes), 120 k points: device {tg * 1e3:.1f} ms, oracle {tc * 1e3:.0f} ms')
