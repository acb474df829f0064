import torch
dev='cuda:0'
torch.manual_seed(0)
m=torch.randint(0,1<<27,(510000,),device=dev,dtype=torch.int32)
m[::3]=m[1::3][:len(m[::3])]   # duplicates
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
a=torch.sort(m,stable=True)[1]; b=torch.sort(m,stable=False)[1]; c=torch.argsort(m.long()*1000000+torch.arange(len(m),device=dev))
print('same perm stable vs not:', torch.equal(a,b), ' stable == (key,index) order:', torch.equal(a,c))
print('stable', round(t(lambda: torch.sort(m,stable=True))), 'us; unstable', round(t(lambda: torch.sort(m,stable=False))), 'us')
