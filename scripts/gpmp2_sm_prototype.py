"""CPU prototype (numpy fp64) behind the Sherman-Morrison form of the GPMP2 solve (csrc/mpb_gpmp2.hip, template flag SM): the block
elimination of a synthetic D = 7 chain with (gj) the collision term assembled into S_t and S_t inverted by unpivoted 2 x 2-block
Gauss-Jordan -- the kernel's form of rounds 1-4 -- and (sm) R_t inverted alone with r_rest and h as extra columns and the rank-1 term
applied by Sherman-Morrison, both against the dense solution refined in long double; `dense chol` = what a dense fp64 Cholesky gives.
    ratio 1e6: gj 5.5e-12 / sm 6.3e-13;  1e8: 2.6e-8 / 5.1e-11;  1e10: 1.9e-4 / 5.9e-9;  1e12: 4.0 / 7.0e-7  (dense chol 8.1e-7)"""
import numpy as np
np.random.seed(0)
D=7; dim=14; H=64
def run(kgp,kc,ks,kg,delta,mode):
    dt=5.0/H
    a=12/dt**3*kgp; bq=-6/dt**2*kgp; cq=4/dt*kgp
    I=np.eye(D)
    Qi=np.block([[a*I,bq*I],[bq*I,cq*I]])
    Phi=np.block([[I,dt*I],[0*I,I]])
    PQP=Phi.T@Qi@Phi
    U=-Phi.T@Qi
    rng=np.random.RandomState(1)
    h=rng.randn(H,D)*0.5; c=np.abs(rng.randn(H))*0.05
    h[0]=0;c[0]=0
    # some waypoints collision free
    free=rng.rand(H)<0.5; h[free]=0; c[free]=0
    x=rng.randn(H,dim)*0.1
    # gradient g (just random-ish consistent): use GP errors
    g=np.zeros((H,dim)); Dg=np.zeros((H,dim,dim))
    for t in range(H):
        Dt=delta*np.eye(dim)
        if t==0: Dt+=ks*np.eye(dim); g[t]+=ks*(0-x[t])
        if t<H-1:
            e=x[t+1]-Phi@x[t]; Dt+=PQP; g[t]+=Phi.T@Qi@e
        if t>0:
            e=x[t]-Phi@x[t-1]; Dt+=Qi; g[t]+=-Qi@e
        if t==H-1: Dt+=kg*np.eye(dim); g[t]+=kg*(0.3-x[t])
        Dg[t]=Dt
    # dense reference in long double w/ refinement
    N=H*dim
    A=np.zeros((N,N)); rhs=np.zeros(N)
    for t in range(H):
        S=Dg[t].copy(); S[:D,:D]+=kc*np.outer(h[t],h[t])
        A[t*dim:(t+1)*dim,t*dim:(t+1)*dim]=S
        r=g[t].copy(); r[:D]+=kc*h[t]*c[t]; rhs[t*dim:(t+1)*dim]=r
        if t<H-1:
            A[t*dim:(t+1)*dim,(t+1)*dim:(t+2)*dim]=U
            A[(t+1)*dim:(t+2)*dim,t*dim:(t+1)*dim]=U.T
    L=np.linalg.cholesky(A)
    import scipy.linalg as sl
    xs=sl.cho_solve((L,True),rhs)
    Al=A.astype(np.longdouble); xl=xs.astype(np.longdouble); rl=rhs.astype(np.longdouble)
    for _ in range(5):
        res=(rl-Al@xl).astype(np.float64)
        xl=xl+sl.cho_solve((L,True),res).astype(np.longdouble)
    ref=xl.astype(np.float64).reshape(H,dim)
    def gj_inv(S):
        # unpivoted 2x2 block gauss-jordan
        n=S.shape[0]; M=S.copy()
        for k0 in range(0,n,2):
            P=M[k0:k0+2,k0:k0+2]; det=P[0,0]*P[1,1]-P[0,1]*P[1,0]
            Pinv=np.array([[P[1,1],-P[0,1]],[-P[1,0],P[0,0]]])/det
            rows=M[k0:k0+2,:].copy(); rows[:,k0:k0+2]=np.eye(2)
            Bm=Pinv@rows
            colK=M[:,k0:k0+2].copy()
            M[:,k0:k0+2]=0
            M=M-colK@Bm
            M[k0:k0+2,:]=Bm
        return M
    W=[None]*H; z=np.zeros((H,dim))
    Sn=np.zeros((dim,dim)); rc=np.zeros(dim)
    for t in range(H):
        R=Dg[t]+Sn; r_rest=g[t]+rc
        if mode=='gj':
            S=R.copy(); S[:D,:D]+=kc*np.outer(h[t],h[t])
            r=r_rest.copy(); r[:D]+=kc*h[t]*c[t]
            Wt=gj_inv(S); zt=Wt@r
        else:
            hp=np.zeros(dim); hp[:D]=h[t]
            aug=np.zeros((dim+2,dim+2)); aug[:dim,:dim]=R; aug[:dim,dim]=r_rest; aug[:dim,dim+1]=hp; aug[dim,dim]=1; aug[dim+1,dim+1]=1
            Ri=gj_inv(aug)
            Rinv=Ri[:dim,:dim]; z0=-Ri[:dim,dim]; y=-Ri[:dim,dim+1]
            # note: GJ inverse of [[R, v],[0,1]] has -R^-1 v in the column; sign flip
            s=hp@y; den=1+kc*s; gfac=kc/den
            Wt=Rinv-gfac*np.outer(y,y)
            zt=z0+gfac*(c[t]-hp@z0)*y
        W[t]=Wt; z[t]=zt
        Sn=-U.T@Wt@U; rc=-U.T@zt
    d=np.zeros((H,dim)); d[H-1]=z[H-1]
    for t in range(H-2,-1,-1):
        d[t]=z[t]-W[t]@(U@d[t+1])
    return np.abs(d-ref).max()/np.abs(ref).max(), np.abs(xs.reshape(H,dim)-ref).max()/np.abs(ref).max()
for sig in [(1e-5,1e-2,1e-5,1e-5),(1e-5,0.1,1e-5,1e-5),(1e-5,1.0,1e-5,1e-5),(1e-5,1.0,1e-5,1e-6),(1e-5,1e-2,1e-5,1e-9)]:
    ks,kgp,kg,kc=[1/s**2 for s in sig]
    for mode in ('gj','sm'):
        e,ed=run(kgp,kc,ks,kg,1e-2,mode)
        print(sig,'ratio %.0e'%((sig[1]/sig[3])**2),mode,'err %.2e'%e,'dense chol %.2e'%ed)
