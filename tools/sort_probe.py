"""Sort / cumsum variants for the Lovasz loss of the workload (19 classes x 1.4 M points): per-class sorts, one batched
sort, flat composite keys -- all ~2 ms; a 2-D cumsum along the long axis is 15x slower than per-class ones."""
import torch
def t(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
N,C=1436991,19
E=torch.rand(C,N,device='cuda')
print("19 x sort(N) f32 desc   : %.2f ms"%t(lambda:[torch.sort(E[c],0,descending=True) for c in range(C)]))
print("sort((C,N),dim=1) desc  : %.2f ms"%t(lambda:torch.sort(E,dim=1,descending=True)))
cls=torch.arange(C,device='cuda',dtype=torch.int64).view(C,1)
def flat():
    k=(cls<<32)|(0xFFFFFFFF-E.view(torch.int32).to(torch.int64))
    return torch.sort(k.view(-1))
print("flat int64 composite    : %.2f ms"%t(flat))
def flat32():
    # 27-bit error key + 5-bit class in int32: lossy (for timing only)
    k=(cls.to(torch.int32)<<26)|(0x3FFFFFF-(E.view(torch.int32)>>4))
    return torch.sort(k.view(-1))
print("flat int32 (timing only): %.2f ms"%t(flat32))
Eh=E.half()
print("19 x sort f16           : %.2f ms"%t(lambda:[torch.sort(Eh[c],0,descending=True) for c in range(C)]))
print("19 x argsort only       : %.2f ms"%t(lambda:[torch.argsort(E[c],descending=True) for c in range(C)]))
print("19 x cumsum(N)          : %.2f ms"%t(lambda:[E[c].cumsum(0) for c in range(C)]))
print("cumsum((C,N),1)         : %.2f ms"%t(lambda:E.cumsum(1)))
