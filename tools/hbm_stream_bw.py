"""Practical streaming bandwidth of this GPU with plain torch ops (copy / add / sum / fill on 445 MB and 1.8 GB tensors):
the yardstick for the HBM-bound kernels of profiles/round2_hbm_ops.md."""
import torch
x=torch.randn(1158864,96,device='cuda'); y=torch.empty_like(x)
def t(fn,n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
b=x.numel()*4
us=t(lambda: y.copy_(x)); print("copy  %.0f us %.0f GB/s"%(us, 2*b/us/1e3))
us=t(lambda: torch.add(x,1.0,out=y)); print("add1  %.0f us %.0f GB/s"%(us, 2*b/us/1e3))
us=t(lambda: x.sum()); print("sum   %.0f us %.0f GB/s"%(us, b/us/1e3))
us=t(lambda: y.fill_(1.0)); print("fill  %.0f us %.0f GB/s"%(us, b/us/1e3))
z=torch.empty_like(x)
us=t(lambda: torch.add(x,y,out=z)); print("add2  %.0f us %.0f GB/s"%(us, 3*b/us/1e3))
x2=torch.randn(1158864*4,96,device='cuda'); y2=torch.empty_like(x2)
us=t(lambda: y2.copy_(x2),20); print("copy 1.8GB %.0f us %.0f GB/s"%(us, 2*x2.numel()*4/us/1e3))
