"""CPU prototype (numpy fp64) of the LOW-RANK form of the GPMP2 solve (round 6; csrc/mpb_gpmp2_lr.hip).

J^T J = A0 + V C V^T:  A0 = GP blocks + priors + damping is the SAME for every particle (the trust-region damping is a batch mean,
gpmp2.py:361-367) and decouples over the degrees of freedom into D chains of 2 x 2 blocks; the collision factors are rank one per
waypoint (V = the h_t embedded at the position rows of waypoint t, C = kc I) and only ACTIVE waypoints (h_t != 0) take part.
    dtheta = A0^-1 (g_rest + V w),   (C^-1 + V^T A0^-1 V) w = c - V^T A0^-1 g_rest
-- `kc c h` never meets A0^-1 on its own (the Sherman-Morrison form of round 5, all waypoints at once).  Per particle: two chain
substitutions with shared factors and one dense SPD solve of the size of its active set; shared: the factors and the
position-position entries G_i(s, t) of A0^-1.  Against the dense solution refined in long double, next to the block elimination
of rounds 1-5 (`gj`) and dense fp64 Cholesky."""
import numpy as np, scipy.linalg as sl, sys
D=7; dim=14
def run(H,kgp,kc,ks,kg,delta,frac_active,seed=1,trust=True,hscale=0.5):
    dt=5.0/H
    a=12/dt**3*kgp; bq=-6/dt**2*kgp; cq=4/dt*kgp
    I=np.eye(D)
    Qi=np.block([[a*I,bq*I],[bq*I,cq*I]]); Phi=np.block([[I,dt*I],[0*I,I]])
    PQP=Phi.T@Qi@Phi; U=-Phi.T@Qi
    rng=np.random.RandomState(seed)
    h=rng.randn(H,D)*hscale; c=np.abs(rng.randn(H))*0.05
    # smooth-ish h along t (neighbouring waypoints push the same way: the near-singular case of the capacitance matrix)
    for t in range(1,H): h[t]=0.9*h[t-1]+0.1*h[t]
    h[0]=0;c[0]=0
    free=np.ones(H,bool); start=rng.randint(1,H//2); n=int(frac_active*H); free[start:start+n]=False   # one contiguous colliding stretch
    extra=rng.rand(H)<0.05; free[extra]=False; free[0]=True
    h[free]=0; c[free]=0
    x=np.cumsum(rng.randn(H,dim)*0.02,0)
    dmean=np.abs(rng.randn(H,dim))*kgp*1e4+1e3      # stand-in for the batch mean of diag(A^T K A)
    g=np.zeros((H,dim)); Dg=np.zeros((H,dim,dim))
    for t in range(H):
        Dt=np.diag(delta*dmean[t]) if trust else delta*np.eye(dim)
        if t==0: Dt=Dt+ks*np.eye(dim); g[t]+=ks*(0-x[t])
        if t<H-1:
            e=x[t+1]-Phi@x[t]; Dt=Dt+PQP; g[t]+=Phi.T@Qi@e
        if t>0:
            e=x[t]-Phi@x[t-1]; Dt=Dt+Qi; g[t]+=-Qi@e
        if t==H-1: Dt=Dt+kg*np.eye(dim); g[t]+=kg*(0.3-x[t])
        Dg[t]=Dt
    N=H*dim
    A0=np.zeros((N,N)); A=np.zeros((N,N)); rhs=np.zeros(N)
    for t in range(H):
        S=Dg[t].copy(); A0[t*dim:(t+1)*dim,t*dim:(t+1)*dim]=S
        S=S.copy(); S[:D,:D]+=kc*np.outer(h[t],h[t]); A[t*dim:(t+1)*dim,t*dim:(t+1)*dim]=S
        r=g[t].copy(); r[:D]+=kc*h[t]*c[t]; rhs[t*dim:(t+1)*dim]=r
        if t<H-1:
            for M in (A,A0):
                M[t*dim:(t+1)*dim,(t+1)*dim:(t+2)*dim]=U; M[(t+1)*dim:(t+2)*dim,t*dim:(t+1)*dim]=U.T
    L=np.linalg.cholesky(A); xs=sl.cho_solve((L,True),rhs)
    Al=A.astype(np.longdouble); xl=xs.astype(np.longdouble); rl=rhs.astype(np.longdouble)
    for _ in range(6):
        res=(rl-Al@xl).astype(np.float64); xl=xl+sl.cho_solve((L,True),res).astype(np.longdouble)
    ref=xl.astype(np.float64)
    # ---- low-rank form, the way the kernel does it: per-dof chains by 2x2 block Thomas (fp64), G table by per-column solves
    def chain(i):
        idx=[ (t*dim+i, t*dim+D+i) for t in range(H)]
        Wt=[None]*H; Ft=[None]*H
        U2=np.array([[U[i,i],U[i,D+i]],[U[D+i,i],U[D+i,D+i]]])
        S=None
        for t in range(H):
            p,v=idx[t]
            Dt=np.array([[A0[p,p],A0[p,v]],[A0[v,p],A0[v,v]]])
            if t>0: Dt=Dt-U2.T@Wt[t-1]@U2
            det=Dt[0,0]*Dt[1,1]-Dt[0,1]*Dt[1,0]
            Wt[t]=np.array([[Dt[1,1],-Dt[0,1]],[-Dt[0,1],Dt[0,0]]])/det
            Ft[t]=Wt[t]@U2
        return Wt,Ft
    chains=[chain(i) for i in range(D)]
    def a0_solve(gv):       # gv (H,dim) -> A0^-1 gv by the chain substitutions
        out=np.zeros((H,dim))
        for i in range(D):
            Wt,Ft=chains[i]
            r=np.zeros((H,2)); z=np.zeros((H,2))
            for t in range(H):
                gt=np.array([gv[t,i],gv[t,D+i]])
                r[t]=gt-(Ft[t-1].T@r[t-1] if t>0 else 0)
                z[t]=Wt[t]@r[t]
            d=np.zeros((H,2)); d[H-1]=z[H-1]
            for t in range(H-2,-1,-1): d[t]=z[t]-Ft[t]@d[t+1]
            out[:,i]=d[:,0]; out[:,D+i]=d[:,1]
        return out
    G=np.zeros((D,H,H))
    for i in range(D):
        for t in range(H):
            e=np.zeros((H,dim)); e[t,i]=1.0
            G[i,:,t]=a0_solve_single(chains[i],e[:,[i,D+i]],H)[:,0] if False else 0
    # (cheaper: all columns of one dof at once)
    for i in range(D):
        Wt,Ft=chains[i]
        E=np.zeros((H,2,H)); E[np.arange(H),0,np.arange(H)]=1.0
        r=np.zeros((H,2,H)); z=np.zeros((H,2,H))
        for t in range(H):
            r[t]=E[t]-(Ft[t-1].T@r[t-1] if t>0 else 0); z[t]=Wt[t]@r[t]
        d=np.zeros((H,2,H)); d[H-1]=z[H-1]
        for t in range(H-2,-1,-1): d[t]=z[t]-Ft[t]@d[t+1]
        G[i]=d[:,0,:]
    act=np.nonzero(np.abs(h).sum(1)>0)[0]; na=len(act)
    u0=a0_solve(g)
    rhs_a=c[act]-np.einsum('ai,ai->a',h[act],u0[act,:D])
    M=np.einsum('ai,bi,iab->ab',h[act],h[act],G[:,act][:,:,act])+np.eye(na)/kc
    Lm=np.linalg.cholesky(M); w=sl.cho_solve((Lm,True),rhs_a)
    g2=g.copy(); g2[act,:D]+=h[act]*w[:,None]
    d=a0_solve(g2).reshape(-1)
    e_lr=np.abs(d-ref).max()/np.abs(ref).max()
    e_ch=np.abs(xs-ref).max()/np.abs(ref).max()
    return na,e_lr,e_ch,np.linalg.cond(M)
if __name__=='__main__':
    for H in (64,128):
        for sig in [(1e-5,1e-2,1e-5,1e-5),(1e-5,0.1,1e-5,1e-5),(1e-5,1.0,1e-5,1e-5),(1e-5,1.0,1e-5,1e-6),(1e-5,1e-2,1e-5,1e-9)]:
            ks,kgp,kg,kc=[1/s**2 for s in sig]
            for fa in (0.1,0.4,0.9):
                for trust in (True,False):
                    na,e,ed,cm=run(H,kgp,kc,ks,kg,1e-2,fa,trust=trust)
                    print('H=%d sig=%s ratio %.0e active %d trust %d: low-rank err %.2e  dense chol %.2e  cond(M) %.1e'%(H,sig,(sig[1]/sig[3])**2,na,trust,e,ed,cm))
