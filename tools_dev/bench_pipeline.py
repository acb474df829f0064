#!/usr/bin/env python
"""Point-pipeline tail: device op vs the host path of the reference's recipe
(scipy cdist + torch masks + randperm per frame) at 16 frames x 20k points, 12 pasted objects."""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from gga_amd import functional as F
from gga_amd import pipelines as P
from gga_amd import synthetic
from gga_amd.points import LiDARPoints
from oracle import oracle as O

B, rng = 16, [0, -40, -3, 70.4, 40, 1]
g = np.random.default_rng(0)
scene = [torch.from_numpy(synthetic.make_pipeline_frame(100 + f, n_points=20000)['points']) for f in range(B)]
centers = [np.stack([g.uniform(2, 68, 12), g.uniform(-38, 38, 12)], 1) for _ in range(B)]
sampled = [torch.from_numpy(np.concatenate([np.concatenate([c + g.normal(0, 0.6, (40, 2)), g.uniform(-2, 0, (40, 1)),
                                                            g.uniform(0, 1, (40, 1))], 1) for c in cs]).astype(np.float32)) for cs in centers]
dev = 'cuda:0'
d_scene = [s.to(dev) for s in scene]; d_samp = [s.to(dev) for s in sampled]
for seeds, tag in (([0] * B, 'no shuffle'), (list(range(1, B + 1)), 'with shuffle')):
    for _ in range(3): prep = F.points_prepare_batch(d_scene, d_samp, centers, 5.0, rng, seeds, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): prep = F.points_prepare_batch(d_scene, d_samp, centers, 5.0, rng, seeds, dev)
    torch.cuda.synchronize()
    print(f'device op ({tag}, inputs resident): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per {B}-frame batch, '
          f'{sum(int(c) for c in prep.counts.cpu())} of {int(prep.capacity_offsets[-1])} rows kept')
t0 = time.perf_counter()
for f in range(B):
    p = P.ObjectSample_GGA.remove_points_in_boxes_v2(LiDARPoints(scene[f], points_dim=4), centers[f], 5.0)
    p = p.cat([LiDARPoints(sampled[f], points_dim=4), p])
    p = p[p.in_range_3d(np.array(rng, np.float32))]
    p.shuffle()
print(f'host path of the recipe (cdist + masks + randperm, 1 thread per frame loop): {(time.perf_counter() - t0) * 1e3:.1f} ms per batch')
t0 = time.perf_counter()
for f in range(B):
    O.points_prepare(scene[f].numpy(), sampled[f].numpy(), centers[f], 5.0, rng)
print(f'oracle (C, no shuffle): {(time.perf_counter() - t0) * 1e3:.1f} ms per batch')
