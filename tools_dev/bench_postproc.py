#!/usr/bin/env python
"""Timing of the inference / pseudo-label kernels at the sizes of the
reference's test_cfg (pre_max_size 4096, max 500 slots), the oracle's CPU time beside them."""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from gga_amd import ops
from gga_amd.pseudo_labels import image_box_match
from oracle import oracle as O
dev = torch.device('cuda:0')


def gpu_time(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def cpu_time(fn, n=1):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3


rng = np.random.default_rng(0)
def rboxes(n, spread=60.0):
    return np.stack([rng.uniform(0, spread, n), rng.uniform(-spread / 2, spread / 2, n), rng.uniform(1.5, 4.5, n),
                     rng.uniform(1.2, 2.2, n), rng.uniform(-3.14, 3.14, n)], 1).astype(np.float32)

print('| op | size | MI355X | oracle (1 CPU thread) |')
print('|---|---|---|---|')
for n in (512, 4096):
    b = rboxes(n); s = rng.uniform(0, 1, n).astype(np.float32)
    tb, ts = torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev)
    g = gpu_time(lambda: ops.nms_rotated(tb, ts, 0.2))
    c = cpu_time(lambda: O.nms_rotated(b, s, 0.2))
    keep = ops.nms_rotated(tb, ts, 0.2)[1]
    print(f'| rotated NMS (thr 0.2, {len(keep)} kept) | {n} boxes | {g:.3f} ms | {c:.1f} ms |')
for n, m in ((500, 500), (4096, 512)):
    a, b = rboxes(n), rboxes(m)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    g = gpu_time(lambda: ops.box_iou_rotated(ta, tb))
    c = cpu_time(lambda: O.box_iou_rotated(a, b))
    print(f'| rotated BEV IoU matrix | {n} x {m} | {g:.3f} ms ({n * m / g / 1e6:.2f} G pairs/s) | {c:.1f} ms |')
B, M, T = 16, 20000, 64
pts = rng.uniform(-40, 70, (B, M, 3)).astype(np.float32)
bx = np.concatenate([rng.uniform(-30, 60, (B, T, 2)), rng.uniform(-2, 0, (B, T, 1)), rng.uniform(1, 5, (B, T, 3)),
                     rng.uniform(-3.14, 3.14, (B, T, 1))], 2).astype(np.float32)
tp, tb = torch.from_numpy(pts).to(dev), torch.from_numpy(bx).to(dev)
g = gpu_time(lambda: ops.points_in_boxes_part(tp, tb))
c = cpu_time(lambda: [O.points_in_boxes(pts[i], bx[i]) for i in range(B)])
print(f'| points_in_boxes_part | {B} x {M} points x {T} boxes | {g:.3f} ms | {c:.1f} ms |')
g = gpu_time(lambda: ops.points_in_boxes_all(tp, tb))
print(f'| points_in_boxes_all | same | {g:.3f} ms ({B * M * T * 4 / g / 1e6:.1f} GB/s of flags written) | - |')
dts, gts = [], []
for f in range(7481):
    ng, nd = int(rng.integers(1, 15)), int(rng.integers(0, 20))
    gg = rng.uniform(0, 1200, (ng, 2)); gg = np.concatenate([gg, gg + rng.uniform(5, 300, (ng, 2))], 1)
    dts.append((gg[rng.integers(0, ng, nd)] + rng.uniform(-20, 20, (nd, 4))).astype(np.float32)); gts.append(gg)
g = gpu_time(lambda: image_box_match(dts, gts), n=3)
c = cpu_time(lambda: O.pseudo_label_match(dts, gts))
print(f'| pseudo-label match (incl. host packing + H2D/D2H) | 7481 frames, {sum(len(d) for d in dts)} detections | {g:.1f} ms | {c:.1f} ms |')
