import torch
dev=torch.device('cuda:0')
a=torch.empty(1<<30,device=dev); b=torch.empty(1<<30,device=dev)
def t(fn,n=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e-3
tz=t(lambda:a.zero_()); print("fill 4 GiB: %.2f TB/s"%(4*(1<<30)/tz/1e12))
tc=t(lambda:b.copy_(a)); print("copy 4 GiB: %.2f TB/s read + same written"%(4*(1<<30)/tc/1e12))
ts=t(lambda:a.sum()); print("read 4 GiB: %.2f TB/s"%(4*(1<<30)/ts/1e12))
