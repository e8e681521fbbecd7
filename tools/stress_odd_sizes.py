import sys, time
sys.path.insert(0, '.')
import numpy as np
from oracle import pg_oracle as O, gc_oracle as G6, sg_oracle as S2, qvi_oracle as Q
from proximalgalerkin_amd import fem, signorini as sg
from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default
from proximalgalerkin_amd.thermoforming import ThermoformingProblem
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
# ex06 rectangular odd mesh: one Newton solve at alpha=1 from zero, compare state
for (N, M) in [(37, 23), (50, 11)]:
    p = GradientConstraintProblem(fem.create_unit_square(N, M), phi_default, f_default)
    c, e = O.create_rectangle(N, M, (0., 0.), (1., 1.)); pr = G6.GradientConstraintP2(c, e)
    x = np.zeros(pr.ntot); xk = x.copy(); res = []
    for i in range(4):
        p.set_alpha(2.0**i); r, its = p.solve()
        xn, rr, it2 = O.newton_solve(pr, x, xk, 2.0**i, O.SnesOptions(rtol=1e-9, atol=1e-9, stol=1e-9, max_it=20))
        res.append((its, it2)); x = xn; xk = x.copy(); p.advance_prev()
    print("ex06", N, M, res, rel(p.get_state()[:pr.n2], x[:pr.n2]), flush=True); p.close()
# ex02 odd box
for n in [(13, 7, 9), (21, 17, 5)]:
    mesh = sg.create_unit_cube(*n); mt, bcs = sg.native_tags(mesh)
    it, its, x, _ = sg.solve_contact_problem(mesh, mt, bcs, verbose=False, return_solution=True, gap=-0.05)
    c, t = S2.create_unit_cube_tets(*n)
    pr = S2.SignoriniP1(c, t, S2.boundary_facets_where(c, t, lambda z: np.isclose(z[:, 2], 0.0)), np.flatnonzero(np.isclose(c[:, 2], 1.0)), gap=-0.05)
    xr, itr, itsr = S2.solve_contact_problem(pr)
    print("ex02", n, its, itsr, rel(x[:3*pr.nv], xr[:3*pr.nv]), flush=True)
# ex01 P2 rectangular via LU
from proximalgalerkin_amd.obstacle import setup_problem, run_outer_loop
for (N, M) in [(27, 14)]:
    msh = fem.create_rectangle(((-1., -1.), (1., 1.)), (N, M))
    pb, sol, solk, al = setup_problem(msh, 2)
    h = run_outer_loop(pb, sol, solk, al, 100, "double_exponential", 1e2, 1e-4)
    c, e = O.create_rectangle(N, M); pr = O.ObstacleLagrange(c, e, 2)
    xr, hr = O.solve_problem(pr, 100, "double_exponential", 1e2, 1e-4)
    print("ex01P2", N, M, h["Newton steps"], hr["Newton steps"], rel(sol.x.array[:pr.n], xr[:pr.n]), flush=True); pb.close()
