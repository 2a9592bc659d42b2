"""Expected node visits / triangle tests (area-weighted) of the host SAH tree over the 262 k-triangle scene with and without spatial
pre-splitting of the triangles whose box exceeds a multiple of the median box (each reference clipped exactly to its cell).  CPU only;
DESIGN.md negative result 38."""
import sys, ctypes as C, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np, make_sponza_class as m
from capsaicin_amd import capi
lib=capi.lib()
lib.cap_host_sah_build.argtypes=[C.c_void_p,C.c_uint32,C.c_void_p,C.c_void_p,C.c_void_p]
V=[]
for name,mat,kind,(p,n,t,tris) in m.build(1.0):
    V.append(p.astype(np.float32)[tris])
V=np.concatenate(V).astype(np.float64)   # (N,3,3)
N=len(V)
def clip_poly(poly, axis, val, keep_less):
    out=[]
    for i in range(len(poly)):
        a,b=poly[i],poly[(i+1)%len(poly)]
        ia=(a[axis]<=val) if keep_less else (a[axis]>=val)
        ib=(b[axis]<=val) if keep_less else (b[axis]>=val)
        if ia: out.append(a)
        if ia!=ib:
            t=(val-a[axis])/(b[axis]-a[axis]); q=a+t*(b-a); q[axis]=val; out.append(q)
    return out
def split_refs(thr_area, max_depth):
    lo=V.min(1); hi=V.max(1); d=hi-lo
    ha=d[:,0]*d[:,1]+d[:,1]*d[:,2]+d[:,2]*d[:,0]
    boxes=[]; nsplit=0
    big=np.nonzero(ha>thr_area)[0]
    small=np.ones(N,bool); small[big]=False
    out_lo=[lo[small]]; out_hi=[hi[small]]
    extra_lo=[];extra_hi=[]
    for i in big:
        stack=[([V[i,0].copy(),V[i,1].copy(),V[i,2].copy()],0)]
        while stack:
            poly,dep=stack.pop()
            P=np.array(poly); l=P.min(0); h=P.max(0); dd=h-l
            a=dd[0]*dd[1]+dd[1]*dd[2]+dd[2]*dd[0]
            if a<=thr_area or dep>=max_depth:
                extra_lo.append(l); extra_hi.append(h); continue
            ax=int(np.argmax(dd)); mid=0.5*(l[ax]+h[ax])
            A=clip_poly(poly,ax,mid,True); B=clip_poly(poly,ax,mid,False)
            nsplit+=1
            if len(A)>=3: stack.append((A,dep+1))
            if len(B)>=3: stack.append((B,dep+1))
    if extra_lo:
        out_lo.append(np.array(extra_lo)); out_hi.append(np.array(extra_hi))
    return np.concatenate(out_lo), np.concatenate(out_hi), nsplit
def metric(lo,hi):
    n=len(lo)
    tb=np.zeros((n,8),np.float32); tb[:,0:3]=lo; tb[:,4:7]=hi
    nodes=np.zeros((n-1,16),np.float32); order=np.zeros(n,np.uint32); depth=C.c_uint32()
    t0=time.time()
    rc=lib.cap_host_sah_build(tb.ctypes.data,n,nodes.ctypes.data,order.ctypes.data,C.byref(depth)); assert rc==0
    bt=time.time()-t0
    bn=nodes.astype(np.float64)
    # node layout: q0 = lo0.xyz hi0.x ; q1 = hi0.yz lo1.xy ; q2 = lo1.z hi1.xyz ; i.e. floats 0..5 child0 lo,hi ; 6..11 child1
    ext=lambda l,h:(h-l)[:,0]*(h-l)[:,1]+(h-l)[:,1]*(h-l)[:,2]+(h-l)[:,2]*(h-l)[:,0]
    a0=ext(bn[:,0:3],bn[:,3:6]); a1=ext(bn[:,6:9],bn[:,9:12])
    kid=nodes[:,12:14].copy().view(np.int32)
    root=ext(np.minimum(bn[:1,0:3],bn[:1,6:9]),np.maximum(bn[:1,3:6],bn[:1,9:12]))[0]
    inner=(a0[kid[:,0]>=0].sum()+a1[kid[:,1]>=0].sum())/root+1.0
    leaf=(a0[kid[:,0]<0].sum()+a1[kid[:,1]<0].sum())/root
    return inner,leaf,depth.value,bt
lo=V.min(1);hi=V.max(1)
d=hi-lo; ha=d[:,0]*d[:,1]+d[:,1]*d[:,2]+d[:,2]*d[:,0]
med=np.median(ha)
print("triangles",N,"median box half-area %.3e"%med)
i,l,dp,bt=metric(lo,hi); print("unsplit: refs %d expected node visits %.2f leaf (triangle) tests %.2f depth %d build %.2fs"%(N,i,l,dp,bt))
for thr,md in ((4*med,2),(2*med,2),(1*med,2),(1*med,4),(0.5*med,3)):
    t0=time.time(); slo,shi,ns=split_refs(thr,md); st=time.time()-t0
    i,l,dp,bt=metric(slo,shi)
    print("split thr %.1fx median depth<=%d: refs %d (+%.0f%%) node visits %.2f triangle tests %.2f depth %d (split %.0fs)"%(thr/med,md,len(slo),100*(len(slo)/N-1),i,l,dp,st))
