import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from gga_amd import functional as F
dev='cuda:0'
B,D,H,W,C=8,2,200,176,128
torch.manual_seed(0)
coors=(torch.rand(B,D,H,W)<0.5).nonzero().int().to(dev)
f=torch.randn(len(coors),C,device=dev,requires_grad=True)
g=torch.randn(B,C*D,H,W,device=dev).contiguous(memory_format=torch.channels_last)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
out=F.sparse_bev_channels_last(f,coors,B,D,H,W)
print(len(coors),'sites; fwd (memset + scatter)',round(t(lambda: F.sparse_bev_channels_last(f,coors,B,D,H,W))),'us; fwd+bwd',round(t(lambda: F.sparse_bev_channels_last(f,coors,B,D,H,W).backward(g))),'us')
