"""Accuracy of the two-piece fp16 split (split_common.hpp) in numpy: one 128 x 128 layer against float64 for several input scales, with and
without flushing subnormals, beside a plain fp32 GEMM and the three-piece bf16 split; then the effect of power-of-two scaling.
   python tools/f16_split_accuracy.py"""
import numpy as np
rng=np.random.default_rng(0)
def split_bf16_3(a):
    # truncation-based 3-way split as 8-bit pieces (round to nearest on each)
    def bf(x):
        u=x.astype(np.float32).view(np.uint32).astype(np.uint64)
        r=((u+0x7FFF+((u>>16)&1))>>16)<<16
        return r.astype(np.uint32).view(np.float32)
    h=bf(a); m=bf(a-h); l=bf(a-h-m); return [h,m,l]
def split_f16_2(a, ftz=False):
    h=a.astype(np.float16).astype(np.float32)
    r=(a-h).astype(np.float32)
    l=r.astype(np.float16).astype(np.float32)
    if ftz:
        l=np.where(np.abs(l)<2.0**-14,0,l); h=np.where(np.abs(h)<2.0**-14,0,h)
    return [h,l]
def mm(pa,pb,pairs):
    acc=np.zeros((pa[0].shape[0],pb[0].shape[1]),np.float64)
    for i,j in pairs:
        acc+=pa[i].astype(np.float64)@pb[j].astype(np.float64)
    return acc.astype(np.float32)   # (accumulate exactly, round once: optimistic about accumulation; isolates representation/dropped terms)
K=128
for scale_x, scale_w in ((1.0,0.1),(0.05,0.1),(1.0,0.01),(30.0,0.3)):
    X=(rng.standard_normal((4096,K))*scale_x).astype(np.float32)
    W=(rng.uniform(-1,1,(K,K))*scale_w*1.5).astype(np.float32)
    ref=X.astype(np.float64)@W.astype(np.float64)
    den=np.abs(ref).max()
    f32=(X@W)   # numpy fp32 (blocked)
    # sequential fp32 fma chain emulation too costly; use numpy's
    e_f32=np.abs(f32-ref).max()/den
    xb=split_bf16_3(X); wb=split_bf16_3(W)
    e_b6=np.abs(mm(xb,wb,[(0,0),(0,1),(1,0),(0,2),(2,0),(1,1)])-ref).max()/den
    for ftz in (False,True):
        xf=split_f16_2(X,ftz); wf=split_f16_2(W,ftz)
        e_f4=np.abs(mm(xf,wf,[(0,0),(0,1),(1,0),(1,1)])-ref).max()/den
        e_f3=np.abs(mm(xf,wf,[(0,0),(0,1),(1,0)])-ref).max()/den
        print(f"sx={scale_x} sw={scale_w} ftz={ftz}: fp32 {e_f32:.2e}  bf16x3/6 {e_b6:.2e}  f16x2/4 {e_f4:.2e}  f16x2/3 {e_f3:.2e}")
print("---- scaled")
def run(X,W,sx,sw,ftz):
    ref=X.astype(np.float64)@W.astype(np.float64); den=np.abs(ref).max()
    xf=split_f16_2(X*np.float32(sx),ftz); wf=split_f16_2(W*np.float32(sw),ftz)
    out=mm(xf,wf,[(0,0),(0,1),(1,0)]).astype(np.float64)/(sx*sw)
    return np.abs(out-ref).max()/den
X=(rng.standard_normal((4096,K))).astype(np.float32)
W=(rng.uniform(-1,1,(K,K))*0.15).astype(np.float32)
# relu-like sparse small activations too
Xr=np.maximum(rng.standard_normal((4096,K))*0.2-0.1,0).astype(np.float32)
for name,XX in (("normal",X),("relu-small",Xr)):
  for lx in (0,4,8,10,12,13):   # activations scaled so that max|x| ~ 2^lx * max
    sx=2.0**lx/ max(1.0,1.0); 
    for lw in (0,8,12,16):
        sw=2.0**lw
        if np.abs(XX).max()*sx>60000 or np.abs(W).max()*sw>60000: continue
        print(name, "sx=2^%d sw=2^%d"%(lx,lw), "ieee %.2e  ftz %.2e"%(run(XX,W,sx,sw,False),run(XX,W,sx,sw,True)))
