"""One-rank sharded solve vs plain solve at 2048^2: cost of the owned-compact Krylov space, gathers and level views\n(no communication).  Measured: 574 ms vs 549 ms."""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from proximalgalerkin_amd import comm as pcomm, fem
from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem
N=2048
for mode in ("plain", "sharded-1rank"):
    c = pcomm.rccl_single(0) if mode != "plain" else None
    msh = fem.create_rectangle(((-1.0,-1.0),(1.0,1.0)), (N,N), comm=c)
    problem, sol, sol_k, alpha = setup_problem(msh, 1)
    for rep in range(3):
        t0=time.perf_counter()
        h = run_outer_loop(problem, sol, sol_k, alpha, 500, "double_exponential", 1e2, 1e-4)
        dt=time.perf_counter()-t0
    print(mode, f"{dt*1e3:.1f} ms", sum(h["Newton steps"]))
    problem.close()
