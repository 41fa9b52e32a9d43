#!/usr/bin/env python3
"""CPU simulation of the NUMERICS of Winograd F(2x2, 3x3) in split precision (f16 pairs hi + lo, three products, f32
accumulation) on the dominant shape of the backbone (14 x 14, 256 -> 256), against a float64 direct convolution — beside
the direct split-precision form the product uses, the float32 direct form, and the plain-f16 forms.  Record:
profiles/experiments/r04_winograd_sp.txt.   python tools/experiments/winograd_sp_numerics.py   (about 20 s on 8 cores)"""
import numpy as np
rng=np.random.default_rng(0)
N,H,W,C,K=4,14,14,256,256
x=rng.standard_normal((N,H,W,C)).astype(np.float32)*1.0
x=np.where(x>0,x,0.25*x)   # prelu-like
w=(rng.standard_normal((K,C,3,3))*np.sqrt(2/(9*C))).astype(np.float64)
def split(a,scale):
    a=a*scale
    hi=a.astype(np.float16); lo=(a-hi.astype(np.float64)).astype(np.float16)
    return hi,lo
def scale_of(a): # pow2 so that max in [1024,2048)
    m=np.abs(a).max(); return 2.0**(10-np.floor(np.log2(m)))
# reference f64 direct
xp=np.pad(x.astype(np.float64),((0,0),(1,1),(1,1),(0,0)))
ref=np.zeros((N,H,W,K))
for ky in range(3):
    for kx in range(3):
        ref+=xp[:,ky:ky+H,kx:kx+W,:]@w[:,:,ky,kx].T
# SP direct: x hi/lo, w hi/lo, 3 products, f32 accumulate
sx=scale_of(x); xh,xl=split(x.astype(np.float64),sx)
sw=scale_of(w); wh,wl=split(w,sw)
def padded(a): return np.pad(a.astype(np.float32),((0,0),(1,1),(1,1),(0,0)))
acc=np.zeros((N,H,W,K),np.float32)
for ky in range(3):
    for kx in range(3):
        for (a,b) in ((xh,wh),(xh,wl),(xl,wh)):
            acc+=padded(a)[:,ky:ky+H,kx:kx+W,:]@b[:,:,ky,kx].astype(np.float32).T
sp=acc.astype(np.float64)/(sx*sw)
print("SP direct  max|d|/max|y| = %.3e  rms %.3e"%(np.abs(sp-ref).max()/np.abs(ref).max(), np.sqrt(((sp-ref)**2).mean())/np.abs(ref).max()))
# Winograd SP
Bt=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64)
G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float64)
At=np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)
U=np.einsum('ai,kcij,bj->abkc',G,w,G)           # 4,4,K,C f64
sU=scale_of(U); Uh,Ul=split(U,sU)
# input: from stored hi+lo (scaled) in f32
xs=(xh.astype(np.float32)+xl.astype(np.float32))  # f32 of scaled input (22 bits -> fits 24)
xsp=np.pad(xs,((0,0),(1,1),(1,1),(0,0)))
tiles=np.zeros((N,7,7,4,4,C),np.float32)
for ty in range(7):
    for tx in range(7):
        tiles[:,ty,tx]=xsp[:,2*ty:2*ty+4,2*tx:2*tx+4,:]
Bt32=Bt.astype(np.float32)
V=np.einsum('ai,ntuijc,bj->ntuabc',Bt32,tiles,Bt32).astype(np.float32)   # f32 exactish (sums of 4 of 22-bit values: exact in f32? up to 2 extra bits -> 24: mostly exact)
Vh=V.astype(np.float16); Vl=(V-Vh.astype(np.float32)).astype(np.float16)
print("V max",np.abs(V).max(),"inf in Vh",np.isinf(Vh).any())
M=np.zeros((N,7,7,4,4,K),np.float32)
for a in range(4):
    for b in range(4):
        for (p_,q_) in ((Vh,Uh),(Vh,Ul),(Vl,Uh)):
            M[:,:,:,a,b,:]+=p_[:,:,:,a,b,:].astype(np.float32)@q_[a,b].astype(np.float32).T
At32=At.astype(np.float32)
Y=np.einsum('ia,ntuabk,jb->ntuijk',At32,M,At32)   # f32
out=np.zeros((N,H,W,K))
for ty in range(7):
    for tx in range(7):
        out[:,2*ty:2*ty+2,2*tx:2*tx+2,:]=Y[:,ty,tx].astype(np.float64)/(sx*sU)
print("Winograd SP max|d|/max|y| = %.3e  rms %.3e"%(np.abs(out-ref).max()/np.abs(ref).max(), np.sqrt(((out-ref)**2).mean())/np.abs(ref).max()))
# f32 direct conv for comparison
acc=np.zeros((N,H,W,K),np.float32)
xp32=np.pad(x,((0,0),(1,1),(1,1),(0,0)))
for ky in range(3):
    for kx in range(3):
        acc+=xp32[:,ky:ky+H,kx:kx+W,:]@w[:,:,ky,kx].astype(np.float32).T
print("f32 direct max|d|/max|y| = %.3e rms %.3e"%(np.abs(acc-ref).max()/np.abs(ref).max(), np.sqrt(((acc-ref)**2).mean())/np.abs(ref).max()))
# plain f16 direct (1 product)
acc=np.zeros((N,H,W,K),np.float32)
for ky in range(3):
    for kx in range(3):
        acc+=padded(xh)[:,ky:ky+H,kx:kx+W,:]@wh[:,:,ky,kx].astype(np.float32).T
f16=acc.astype(np.float64)/(sx*sw)
print("f16 direct max|d|/max|y| = %.3e rms %.3e"%(np.abs(f16-ref).max()/np.abs(ref).max(), np.sqrt(((f16-ref)**2).mean())/np.abs(ref).max()))
# winograd plain f16
M=np.zeros((N,7,7,4,4,K),np.float32)
for a in range(4):
    for b in range(4):
        M[:,:,:,a,b,:]+=Vh[:,:,:,a,b,:].astype(np.float32)@Uh[a,b].astype(np.float32).T
Y=np.einsum('ia,ntuabk,jb->ntuijk',At32,M,At32)
for ty in range(7):
    for tx in range(7):
        out[:,2*ty:2*ty+2,2*tx:2*tx+2,:]=Y[:,ty,tx].astype(np.float64)/(sx*sU)
print("Winograd f16 max|d|/max|y| = %.3e rms %.3e"%(np.abs(out-ref).max()/np.abs(ref).max(), np.sqrt(((out-ref)**2).mean())/np.abs(ref).max()))
