import sys; sys.path.insert(0,'.')
import numpy as np
from scipy.optimize import brentq
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.obstacle import solve_problem
r0=0.5; a=brentq(lambda a:a*a*(1-np.log(a))-r0*r0,0.1,0.45); c=a*a/np.sqrt(r0*r0-a*a)
def disk(h,curved):
    m=fem.create_disk(h); e,ce=m.edges(); mid=0.5*(m.geometry[e[:,0]]+m.geometry[e[:,1]])
    if curved:
        b=np.flatnonzero(np.bincount(ce.ravel(),minlength=len(e))==1); mid[b]/=np.linalg.norm(mid[b],axis=1)[:,None]
    return fem.Mesh(m.geometry,m.cells,midside=mid)
for h in (0.2,0.1,0.05):
    out=[]
    for curved in (False,True):
        mesh=disk(h,curved)
        sol,newton,hist=solve_problem(mesh,2,100,"constant",1e5,1e-8,verbose=False,return_history=True)
        nv=mesh.num_vertices; r=np.hypot(mesh.geometry[:,0],mesh.geometry[:,1]); o=r>=0.6
        exact=np.where(r<=a,np.sqrt(np.maximum(r0*r0-r*r,0)),-c*np.log(np.maximum(r,1e-300)))
        out.append((np.abs(sol.x.array[:nv][o]-exact[o]).max(), np.abs(sol.x.array[:nv]-exact).max(), sum(hist["Newton steps"])))
    print(f"h={h}: polygon outer err {out[0][0]:.3e} all {out[0][1]:.3e} newton {out[0][2]} | curved outer err {out[1][0]:.3e} all {out[1][1]:.3e} newton {out[1][2]}")
