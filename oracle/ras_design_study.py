"""Design study (numpy) for the multi-GPU preconditioner: two-level restricted additive Schwarz over horizontal
strips with inexact local V-cycles + a replicated global coarse V-cycle.  Numbers quoted in DESIGN.md section 7.
TEST INFRASTRUCTURE / PROTOTYPE ONLY; needs /tmp/systems_N_B.npz produced from oracle runs (see git history)."""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from oracle import pg_oracle as O, krylov_proto as KP

class RectMG:
    """collective-smoother MG on an nx x ny right-diagonal rectangle given K,M,D (scalar CSR) and u-mask; optional psi mask"""
    def __init__(self, K, Mm, D, alpha, nx, ny, mask_u, mask_p=None, nu=2, omega=0.8, min_level=2, start_only=False):
        self.nu, self.omega = nu, omega; self.levels=[]
        mask_p = np.zeros_like(mask_u) if mask_p is None else mask_p
        K,Mm,D=K.tocsr(),Mm.tocsr(),D.tocsr()
        while True:
            ku=(~mask_u).astype(float); kp=(~mask_p).astype(float)
            A=(sp.diags(ku)@(alpha*K)@sp.diags(ku)+sp.diags(1-ku)).tocsr()
            B=(sp.diags(ku)@Mm@sp.diags(kp)).tocsr()
            Dm=(sp.diags(kp)@D@sp.diags(kp)+sp.diags(1-kp)*(-1.0)).tocsr()   # psi-dirichlet rows: -D -> identity (so -(-1)=1)
            L=dict(A=A,B=B,BT=B.T.tocsr(),D=Dm,nx=nx,ny=ny,mask_u=mask_u,mask_p=mask_p)
            a,b,d=A.diagonal(),B.diagonal(),Dm.diagonal(); L['blk']=(a,b,d,-(a*d)-b*b)
            self.levels.append(L)
            if nx<=min_level or ny<=min_level or nx%2 or ny%2: break
            P=KP.interp_matrix(nx,ny); L['P']=P
            K,Mm,D=(P.T@K@P).tocsr(),(P.T@Mm@P).tocsr(),(P.T@D@P).tocsr()
            sxf=nx+1; nx//=2; ny//=2
            I,J=np.meshgrid(np.arange(nx+1),np.arange(ny+1),indexing='xy'); fidx=(2*J*sxf+2*I).ravel()
            mask_u=mask_u[fidx]; mask_p=mask_p[fidx]
    def _apply(self,L,xu,xp): return L['A']@xu+L['B']@xp, L['BT']@xu-L['D']@xp
    def _smooth(self,L,xu,xp,ru,rp,its):
        a,b,d,det=L['blk']; omu=np.where(L['mask_u'],1.0,self.omega); omp=np.where(L['mask_p'],1.0,self.omega)
        for _ in range(its):
            yu,yp=self._apply(L,xu,xp); su,s_p=ru-yu,rp-yp
            xu=xu+omu*(-d*su-b*s_p)/det; xp=xp+omp*(-b*su+a*s_p)/det
        return xu,xp
    def vcycle(self,ru,rp,l=0):
        L=self.levels[l]; xu,xp=np.zeros_like(ru),np.zeros_like(rp)
        if 'P' not in L: return self._smooth(L,xu,xp,ru,rp,8)
        xu,xp=self._smooth(L,xu,xp,ru,rp,self.nu)
        yu,yp=self._apply(L,xu,xp); C=self.levels[l+1]
        cu,cp=self.vcycle((~C['mask_u'])*(L['P'].T@(ru-yu)),(~C['mask_p'])*(L['P'].T@(rp-yp)),l+1)
        xu,xp=xu+L['P']@cu,xp+L['P']@cp
        return self._smooth(L,xu,xp,ru,rp,self.nu)

N=int(sys.argv[1]); G=int(sys.argv[2]); delta=int(sys.argv[3]); Lc=int(sys.argv[4]); psiD=int(sys.argv[5])
d=np.load(f"/tmp/systems_{N}_B.npz"); xs=d['xs']; alphas=d['alphas']; xks=d['xks']
c,ce=O.create_rectangle(N,N); P=O.ObstacleP1(c,ce,O.boundary_vertices_rectangle(N,N)); n=P.n; sx=N+1
rows=np.arange(n)//sx
for k in (0,8,11,16,len(xs)-1):
    x=xs[k]; a=alphas[k]; J=P.jacobian(x,a).tocsr(); b=-P.residual(x,xks[k],a)
    D=sp.csr_matrix((P.jacobian_blocks(x),P.indices_s,P.indptr_s),shape=(n,n))
    # global hierarchy (for coarse correction from level Lc) -- reference single-domain MG too
    mg_glob=RectMG(P.K,P.M,D,a,N,N,P.isbc)
    # transfer chain fine->Lc
    Pchain=None
    for l in range(Lc):
        Pl=mg_glob.levels[l]['P']; Pchain=Pl if Pchain is None else Pchain@Pl
    class Sub: pass
    subs=[]; bounds=np.linspace(0,N+1,G+1).astype(int)
    for g in range(G):
        a0,b0=bounds[g],bounds[g+1]
        lo=max(a0-delta,0); hi=min(b0+delta,N+1)
        # make local cell-row count even-friendly: not enforced
        loc=np.flatnonzero((rows>=lo)&(rows<hi)); nyl=hi-lo-1
        Kl=P.K[loc][:,loc]; Ml=P.M[loc][:,loc]; Dl=D[loc][:,loc]
        art=np.zeros(len(loc),bool)
        if lo>0: art|=(rows[loc]==lo)
        if hi<N+1: art|=(rows[loc]==hi-1)
        mu=P.isbc[loc]|art; mp=art if psiD else np.zeros_like(art)
        s=Sub(); s.loc=loc; s.mg=RectMG(Kl,Ml,Dl,a,N,nyl,mu,mp); s.own=(rows[loc]>=a0)&(rows[loc]<b0); subs.append(s)
    def ras(r):
        z=np.zeros_like(r)
        for s in subs:
            zu,zp=s.mg.vcycle(r[s.loc],r[n+s.loc])
            z[s.loc[s.own]]=zu[s.own]; z[n+s.loc[s.own]]=zp[s.own]
        return z
    def coarse(r):
        ru=Pchain.T@r[:n]; rp=Pchain.T@r[n:]; C=mg_glob.levels[Lc]
        cu,cp=mg_glob.vcycle((~C['mask_u'])*ru,rp,Lc)
        return np.concatenate([Pchain@cu,Pchain@cp])
    def prec_add(r): return ras(r)+coarse(r)
    def prec_mult(r):
        z1=coarse(r); return z1+ras(r-J@z1)
    def prec_single(r): return np.concatenate(mg_glob.vcycle(r[:n],r[n:]))
    out=[]
    for name,pr in (("single",prec_single),("ras",ras),("add",prec_add),("mult",prec_mult)):
        sol,its,h=KP.fgmres(J,b,pr,1e-9,150); out.append(f"{name}:{its}")
    print(k," ".join(out),flush=True)
