"""Per-level sparse conv micro-benchmark on the SECOND config: rows, neighbour density, offsets a
mask-sorted tile computes (128-row and 32-row granularity), kernel time and fp32 rate."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gga_amd import Config, build_model, synthetic, _lib
from gga_amd import functional as F
from gga_amd.sparse import SparseConvTensor
DEV = 'cuda:0'
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs/gga/gga_kitti_config.py'))
torch.manual_seed(0)
model = build_model(cfg.model).to(DEV).train()
b = synthetic.make_batch(BS, n_points=20000, pc_range=synthetic.RANGE_SECOND)
pts = [p.to(DEV) for p in b['points']]
v, n, c = model.voxelize(pts)
f = model.pts_voxel_encoder(v, n, c)
enc = model.pts_middle_encoder
L = _lib.lib()
def unions(mask_sorted, g):
    t = mask_sorted[: (len(mask_sorted) // g) * g].view(-1, g).long()
    u = torch.zeros(t.shape[0], dtype=torch.long, device=t.device)
    for i in range(g): u |= t[:, i]
    return float(sum(((u >> k) & 1) for k in range(27)).float().mean())
with torch.no_grad():
    x = SparseConvTensor(f, c.int(), enc.sparse_shape, BS)
    x = enc.conv_input(x)
    for i, layer in enumerate(enc.encoder_layers):
        for m in layer: x = m(x)
        lv = x._level
        rb = lv.subm_rulebook((3, 3, 3))
        C_ = x.features.shape[1]
        valid = float((rb.nbr >= 0).float().sum(0).mean())
        ms = rb.mask[rb.perm.long()] & 0x7FFFFFF
        u128, u32 = unions(ms, 128), unions(ms, 32)
        w = torch.randn(27, C_, C_, device=DEV) * 0.05
        y = torch.empty(lv.n, C_, device=DEV)
        from gga_amd.sparse import _pack_weight
        wp = _pack_weight(w, 27, C_, C_, 0, split=False)
        feats = x.features.contiguous()
        t = timeit(lambda: L.gga_sparse_conv_apply(F._p(feats), F._p(rb.nbr), F._p(wp), F._p(rb.perm), F._p(rb.mask), lv.n, 27, C_, C_, 0, F._p(y), F._stream()))
        fl = lambda k: 2.0 * lv.n * k * C_ * C_
        wps = _pack_weight(w, 27, C_, C_, 0, split=True)
        y2 = torch.empty_like(y)
        t_x9 = timeit(lambda: L.gga_sparse_conv_apply_split(F._p(feats), F._p(rb.nbr), F._p(wps), F._p(rb.perm), F._p(rb.mask), lv.n, 27, C_, C_, 0, F._p(y2), F._stream()))
        err = float((y2 - y).abs().max() / y.abs().max())
        for nsub in (16384, 65536):
            sel = rb.perm.long()[-nsub:]
            nbr_s = rb.nbr[:, sel].contiguous(); mask_s = rb.mask[sel].contiguous(); ys = torch.empty(nsub, C_, device=DEV)
            t_s = timeit(lambda: L.gga_sparse_conv_apply_split(F._p(feats), F._p(nbr_s), F._p(wps), None, F._p(mask_s), nsub, 27, C_, C_, 0, F._p(ys), F._stream()))
            pcs = float(sum(((mask_s.long() >> kb) & 1) for kb in range(27)).float().mean())
            print(f'   bf16x9 on the last {nsub} rows of the mask order ({nsub // 256} tiles, {pcs:.1f} offsets/row): {t_s:.0f} us')
        print(f'   bf16x9 apply: {t_x9:.0f} us = {fl(u128)/t_x9/1e6:.1f} TF/s computed; max |diff| vs fp32 MFMA / max|y| = {err:.2e}')
        gw = torch.empty_like(w)
        tw = timeit(lambda: L.gga_sparse_conv_wgrad(F._p(feats), F._p(y), F._p(rb.nbr), lv.n, 27, C_, C_, F._p(gw), F._stream()))
        fl = lambda k: 2.0 * lv.n * k * C_ * C_
        gw2 = torch.empty_like(w)
        ws = torch.empty(L.gga_sparse_conv_wgrad_workspace_bytes(lv.n, 27, C_, C_), dtype=torch.uint8, device=DEV)
        tw2 = timeit(lambda: L.gga_sparse_conv_wgrad_split(F._p(feats), F._p(y), F._p(rb.nbr), lv.n, 27, C_, C_, F._p(gw2), F._p(ws), ws.numel(), F._stream()))
        gw3 = torch.empty_like(w)
        L.gga_sparse_conv_wgrad_split(F._p(feats), F._p(y), F._p(rb.nbr), lv.n, 27, C_, C_, F._p(gw3), F._p(ws), ws.numel(), F._stream())
        print(f'   wgrad bf16 planes (deterministic): {tw2:.0f} us = {fl(valid)/tw2/1e6:.1f} TF/s useful; max |diff| vs fp32 MFMA / max = '
              f'{float((gw2 - gw).abs().max() / gw.abs().max()):.2e}; run-to-run bit-identical: {bool(torch.equal(gw2, gw3))}')
        print(f'stage {i+1}: n={lv.n} C={C_} valid/row {valid:.2f} union128 {u128:.2f} union32 {u32:.2f} | apply {t:.0f} us '
              f'= {fl(u128)/t/1e6:.1f} TF/s computed, {fl(valid)/t/1e6:.1f} TF/s useful | wgrad {tw:.0f} us = {fl(27)/tw/1e6:.1f} TF/s dense-equiv, {fl(valid)/tw/1e6:.1f} useful')
