import torch, time
dev='cuda:0'
n=510000
mask=torch.randint(0,2**27,(n,),dtype=torch.int32,device=dev)
key64=torch.randint(0,2**50,(n,),dtype=torch.int64,device=dev)
key32=torch.randint(0,2**31-1,(n,),dtype=torch.int32,device=dev)
def t(name, f):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); print(f'{name:50s} {(time.perf_counter()-t0)/20*1e6:8.1f} us')
t('sort int32 stable', lambda: torch.sort(mask, stable=True))
t('sort int32', lambda: torch.sort(mask))
t('argsort int64', lambda: torch.argsort(key64))
t('argsort int64 stable', lambda: torch.argsort(key64, stable=True))
t('argsort int32', lambda: torch.argsort(key32))
t('sort int64 of (mask<<20|row)', lambda: torch.sort((mask.long()<<20)|torch.arange(n,device=dev)))
t('unique int64 inverse', lambda: torch.unique(key64, return_inverse=True))
