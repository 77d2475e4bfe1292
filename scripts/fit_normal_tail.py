import numpy as np
from scipy.special import erfcx, erf
np.set_printoptions(precision=10)
U=6.0
u=np.cos(np.linspace(0,np.pi,4001))*U/2+U/2
u=np.sort(u)
R=0.5*erfcx(u/np.sqrt(2))
wt=np.maximum(u,0.3)*np.exp(-u*u/2)
for deg in (6,7,8,9,10):
    w=wt.copy()
    for it in range(60):
        V=np.vander(u,deg+1,increasing=True)
        c,_,_,_=np.linalg.lstsq(V*w[:,None],R*w,rcond=None)
        err=np.abs(V@c-R)*wt
        w=w*(1+2*err/err.max())/2; w/=w.max()
    # fp32 evaluation of gelu
    x=np.linspace(-8,8,400001).astype(np.float32)
    a=np.minimum(np.abs(x),np.float32(U))
    c32=c.astype(np.float32)
    p=np.full_like(a,c32[-1])
    for k in range(deg-1,-1,-1): p=p*a+c32[k]
    e=np.exp2((a*a*np.float32(-0.5*np.log2(np.e))).astype(np.float32)).astype(np.float32)
    q=(p*e).astype(np.float32)
    phi=np.where(x<0,q,np.float32(1)-q)
    y=(x*phi).astype(np.float32)
    xr=x.astype(np.float64)
    phir=0.5*(1+erf(xr/np.sqrt(2)))
    yr=xr*phir
    dg=phi+x*np.float32(0.3989422804014327)*e
    dgr=phir+xr*np.exp(-xr*xr/2)/np.sqrt(2*np.pi)
    print(deg,"max w-err",err.max(),"gelu abs err",np.abs(y-yr).max(),"phi err",np.abs(phi-phir).max(),"dgelu err",np.abs(dg-dgr).max())
    print("  coef",repr(c))
