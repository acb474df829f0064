"""Forward time of the SubM 128 -> 128 convolution at the 128-channel level of the shipped config (bs 8) with whatever
library build is loaded (tools_dev/run_with_lib.py + tools_dev/exp_libs/libgga_x9_*.so: ablations of sp_conv_x9_kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, synthetic, _lib, dense_conv
from gga_amd import functional as F
from gga_amd.sparse import SparseConvTensor, _pack_weight
DEV = 'cuda:0'
dense_conv.PLANES = 2
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
cfg = Config.fromfile(os.path.join(root, 'configs/gga/gga_kitti_config.py'))
torch.manual_seed(0)
model = build_model(cfg.model).to(DEV).train()
b = synthetic.make_batch(8, n_points=20000, pc_range=synthetic.RANGE_SECOND)
v, n, c = model.voxelize([p.to(DEV) for p in b['points']])
f = model.pts_voxel_encoder(v, n, c)
enc = model.pts_middle_encoder
L = _lib.lib()
with torch.no_grad():
    x = SparseConvTensor(f, c.int(), enc.sparse_shape, 8)
    x = enc.conv_input(x)
    for layer in enc.encoder_layers:
        for m in layer:
            x = m(x)
        lvl, C_ = x._level, x.features.shape[1]
        if C_ < 64:
            continue
        rb = lvl.subm_rulebook((3, 3, 3))
        feats = torch.randn(lvl.n, C_, device=DEV)
        w = torch.randn(27, C_, C_, device=DEV) * 0.05
        xa, wa = dense_conv._amax_bits(feats), dense_conv._amax_bits(w)
        wp = _pack_weight(w, 27, C_, C_, 0, w_amax=wa)
        y = torch.empty(lvl.n, C_, device=DEV)
        fn = lambda: L.gga_sparse_conv_apply_planes(F._p(feats), F._p(rb.nbr), F._p(wp), F._p(rb.perm), F._p(rb.mask), lvl.n, 27, C_, C_, 0,
                                                    F._p(y), C_, 2, F._p(xa), F._p(wa), F._stream())
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f'{os.path.basename(_lib.LIB_PATH):28s} rows {lvl.n} C {C_}: forward {e0.elapsed_time(e1) * 100:.1f} us', flush=True)
