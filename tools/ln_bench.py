import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from w2v2_speaker_amd import ops, _lib
dev="cuda"; M,H=66*149,768
x=torch.randn(M,H,device=dev).half(); r=torch.randn(M,H,device=dev).half(); y=torch.empty_like(x)
g=torch.ones(H,device=dev); b=torch.zeros(H,device=dev); mean=torch.empty(M,device=dev); rstd=torch.empty(M,device=dev)
dy=torch.randn(M,H,device=dev).half(); ds=torch.empty_like(x); dr=torch.empty_like(x); dg=torch.zeros(H,device=dev); db=torch.zeros(H,device=dev)
def t(fn,reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/reps
print("ln_fwd p=0.1  %.1f us"%t(lambda: ops.layernorm_fwd(x,r,g,b,y,mean,rstd,1e-5,0.1,3)))
print("ln_bwd ws     %.1f us"%t(lambda: ops.layernorm_bwd(dy,r,mean,rstd,g,ds,dr,dg,db,0.1,3)))
L=_lib.load()
def raw():
    _lib.check(L.w2v2_layernorm_bwd(dy.data_ptr(),r.data_ptr(),mean.data_ptr(),rstd.data_ptr(),g.data_ptr(),ds.data_ptr(),dr.data_ptr(),dg.data_ptr(),db.data_ptr(),None,M,H,0.1,3,1,torch.cuda.current_stream().cuda_stream))
print("ln_bwd atomics %.1f us"%t(raw))
def nog():
    _lib.check(L.w2v2_layernorm_bwd(dy.data_ptr(),r.data_ptr(),mean.data_ptr(),rstd.data_ptr(),g.data_ptr(),ds.data_ptr(),dr.data_ptr(),None,None,None,M,H,0.1,3,1,torch.cuda.current_stream().cuda_stream))
print("ln_bwd no dgamma %.1f us"%t(nog))
