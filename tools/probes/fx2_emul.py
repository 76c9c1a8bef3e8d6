import numpy as np, sys
sys.path.insert(0,'/root/repo')
from fakebob_amd.models import synthetic_ubm_moments
rng=np.random.default_rng(0)
C,D,F=2048,72,400
w,mu,var=synthetic_ubm_moments(C,D,2001)
mu=mu.astype(np.float64); var=var.astype(np.float64)
l=(mu/var).astype(np.float32); q=(-0.5/var).astype(np.float32)
gc=(np.log(w)-0.5*(D*np.log(2*np.pi)+np.log(var).sum(1)+(mu*mu/var).sum(1))).astype(np.float32)
sd=np.r_[np.full(24,3.0),np.full(24,1.0),np.full(24,0.5)]
x=(rng.normal(size=(F,D))*sd).astype(np.float32)
x2=(x*x).astype(np.float32)
def split16(v, scale=4096.0):
    v=v.astype(np.float32)
    a=v.astype(np.float16).astype(np.float32)
    r=(v-a)*np.float32(scale)
    b=r.astype(np.float16).astype(np.float32)
    return a.astype(np.float64), b.astype(np.float64)
def ll_exact(l,q,gc,x,x2):
    return gc[None,:].astype(np.float64)+x.astype(np.float64)@l.astype(np.float64).T+x2.astype(np.float64)@q.astype(np.float64).T
ref=ll_exact(l,q,gc,x,x2)
l1,l2=split16(l); q1,q2=split16(q); g1,g2=split16(gc); x1,x2_=split16(x); y1,y2=split16(x2)
hi = x1@l1.T + y1@q1.T + g1[None,:]
mid= x1@l2.T + x2_@l1.T + y1@q2.T + y2@q1.T + g2[None,:]
ap3 = hi+mid/4096.0
ap4 = ap3 + (x2_@l2.T + y2@q2.T)/4096.0**2
# f32 sequential accumulate emulation of error scale: compare with f32 matmul
f32 = (gc[None,:]+x@l.T+x2@q.T).astype(np.float64)
def lse(a): 
    m=a.max(1,keepdims=True); return (m+np.log(np.exp(a-m).sum(1,keepdims=True)))[:,0]
for name,ap in (("f16x2 3prod",ap3),("f16x2 4prod",ap4),("numpy f32 gemm",f32)):
    e=ap-ref
    # only components that matter
    print(name,"max|err| ll:",np.abs(e).max(),"rms:",np.sqrt((e**2).mean()),"lse err max:",np.abs(lse(ap)-lse(ref)).max(),"mean-frame err:",abs((lse(ap)-lse(ref)).mean()))
print("ll magnitude", np.abs(ref).mean(), "lse mean", lse(ref).mean())
