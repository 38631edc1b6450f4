import numpy as np, math
f32=np.float32
def series_s(u):   # (sin r / r - 1)/u = sum_{k>=1} (-1)^k u^(k-1)/(2k+1)!
    return sum(((-1)**k)*u**(k-1)/math.factorial(2*k+1) for k in range(1,14))
def series_c(u):   # (cos r - 1 + u/2)/u^2 = sum_{k>=2} (-1)^k u^(k-2)/(2k)!
    return sum(((-1)**k)*u**(k-2)/math.factorial(2*k) for k in range(2,15))
def fit(func, deg, umax, n=3000):
    k=np.arange(n); u=0.5*umax*(1-np.cos(np.pi*(k+0.5)/n))
    y=func(u); V=np.vander(u/umax, deg+1, increasing=True); w=np.ones(n)
    for it in range(200):
        c=np.linalg.lstsq(V*w[:,None], y*w, rcond=None)[0]
        e=np.abs(V@c-y); w=w*(1+0.5*e/e.max()); w/=w.max()
    return c/umax**np.arange(deg+1)
umax=(np.pi/2*1.0001)**2
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
x=np.linspace(-np.pi/2,np.pi/2,2000001); r=x.astype(f32); u=(r*r).astype(f32)
def ev(cs,cc):
    p=np.full_like(u,f32(cs[-1]))
    for k in range(len(cs)-2,-1,-1): p=fma(u,p,np.full_like(u,f32(cs[k])))
    s=fma((r*u).astype(f32),p,r)
    q=np.full_like(u,f32(cc[-1]))
    for k in range(len(cc)-2,-1,-1): q=fma(u,q,np.full_like(u,f32(cc[k])))
    c=fma((u*u).astype(f32),q,fma(np.full_like(u,f32(-0.5)),u,np.full_like(u,f32(1))))
    return np.abs(s.astype(np.float64)-np.sin(r.astype(np.float64))).max(), np.abs(c.astype(np.float64)-np.cos(r.astype(np.float64))).max()
for ds,dc in ((4,4),(3,4),(3,3),(4,3)):
    cs=fit(series_s,ds,umax); cc=fit(series_c,dc,umax)
    es,ec=ev(cs,cc)
    print(ds,dc,'sin err %.3e cos err %.3e'%(es,ec))
    print(' sin', ', '.join('%.10ef'%float(f32(v)) for v in cs)); print(' cos', ', '.join('%.10ef'%float(f32(v)) for v in cc))

# full functions incl. range reduction, x in [-20, 20]
def fmaf(a,b,c): return (np.float64(1)*a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
X=(np.random.RandomState(0).uniform(-20,20,4000001)).astype(f32)
def F(v): return np.full_like(X,f32(v))
def old(x):
    k=np.rint((x*f32(0.6366197466850281)).astype(f32)).astype(f32)
    r=fmaf(-k,F(1.5707963705062866),x); r=fmaf(-k,F(-4.371138828673793e-08),r); r=fmaf(-k,F(-1.7763568394002505e-15),r)
    r2=(r*r).astype(f32)
    s=fmaf((r*r2).astype(f32), fmaf(r2, fmaf(r2,F(-1.9515295891e-4),F(8.3321608736e-3)),F(-1.6666654611e-1)), r)
    c=fmaf((r2*r2).astype(f32), fmaf(r2, fmaf(r2,F(2.443315711809948e-5),F(-1.388731625493765e-3)),F(4.166664568298827e-2)), fmaf(F(-0.5),r2,F(1.0)))
    q=k.astype(np.int64)
    a=np.where(q&1,c,s); b=np.where(q&1,s,c)
    return np.where(q&2,-a,a), np.where((q+1)&2,-b,b)
pi=np.float64(np.pi); p1=f32(pi); p2=f32(pi-np.float64(p1)); p3=f32(pi-np.float64(p1)-np.float64(p2))
print('pi parts',repr(float(p1)),repr(float(p2)),repr(float(p3)))
cs=fit(series_s,4,umax); cc=fit(series_c,3,umax)
def new(x):
    k=np.rint((x*f32(1/np.pi)).astype(f32)).astype(f32)
    r=fmaf(-k,F(p1),x); r=fmaf(-k,F(p2),r); r=fmaf(-k,F(p3),r)
    u_=(r*r).astype(f32)
    p=F(cs[-1])
    for kk in range(len(cs)-2,-1,-1): p=fmaf(u_,p,F(cs[kk]))
    s=fmaf((r*u_).astype(f32),p,r)
    q=F(cc[-1])
    for kk in range(len(cc)-2,-1,-1): q=fmaf(u_,q,F(cc[kk]))
    c=fmaf((u_*u_).astype(f32),q,fmaf(F(-0.5),u_,F(1.0)))
    sg=np.where(k.astype(np.int64)&1,-1.0,1.0).astype(f32)
    return s*sg, c*sg
for name,fn in (('old',old),('new',new)):
    s,c=fn(X)
    es=np.abs(s.astype(np.float64)-np.sin(X.astype(np.float64))); ec=np.abs(c.astype(np.float64)-np.cos(X.astype(np.float64)))
    print(name,'sin max %.3e rms %.3e   cos max %.3e rms %.3e'%(es.max(),np.sqrt((es**2).mean()),ec.max(),np.sqrt((ec**2).mean())))
