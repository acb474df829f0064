"""Small-grid SparseEncoder under both plane forms against the float64 dense restatement: per-convolution forward error and
per-parameter gradient error (debug aid for test_sparse_encoder_gga_config_vs_dense_reference)."""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
from gga_amd import dense_conv
from gga_amd.sparse import SparseConvolution
from gga_amd.sparse_encoder import SparseEncoder
from oracle import sparse_ref as SR
from test_sparse_gpu import _coords, _sorted_rows
DEV = 'cuda:0'
torch.manual_seed(0)
shape, B = (41, 40, 32), 2
enc = SparseEncoder(in_channels=4, sparse_shape=list(shape), output_channels=128, order=('conv', 'norm', 'act'),
                    encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                    encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')
enc.train()
ref64 = copy.deepcopy(enc).double()
coors = _coords(B, shape, 900, seed=5)
feats = torch.randn(len(coors), 4)
trace = []
y64, _ = SR.sparse_encoder_reference(ref64, feats.double(), coors, B, trace=trace)
g = torch.randn(y64.shape)
y64.backward(g.double())
for planes in (2, 3):
    for fused in (True, False):
        dense_conv.PLANES = planes
        dense_conv.BN_BWD_FUSED = fused
        e = copy.deepcopy(enc).to(DEV)
        got = []
        hooks = [m.register_forward_hook(lambda mod, inp, out: got.append((out.features.detach().cpu(), out.indices.cpu(), tuple(out.spatial_shape))))
                 for m in e.modules() if isinstance(m, SparseConvolution)]
        y = e(feats.to(DEV), coors.to(DEV), B)
        y.backward(g.to(DEV))
        print(f'--- planes {planes} bn_bwd_fused {fused}: out err {float((y.detach().cpu().double() - y64.detach()).norm() / y64.norm()):.2e}')
        for i, ((f1, c1, s1), (f2, c2, s2)) in enumerate(zip(got, trace)):
            a, _ = _sorted_rows(f1, c1, s1)
            b, _ = _sorted_rows(f2.detach(), c2, s2)
            print(f'   conv {i:2d} rows {len(a):5d} C {a.shape[1]:3d} fwd err {float((a.double() - b).norm() / b.norm()):.2e}')
        for (n, p), (_, q) in zip(e.named_parameters(), ref64.named_parameters()):
            print(f'   grad {n:45s} {float((p.grad.cpu().double() - q.grad).norm() / q.grad.norm()):.2e}')
