"""Example 01 driver: the proximal-point outer loop of
/root/reference/examples/01_obstacle_problem/obstacle_pg.py:53-264 on top of the HIP backend.

`solve_problem` keeps the reference's argument order and return value `(sol, sum(Newton_steps))`;
the mesh file name becomes a Mesh object (XDMF/HDF5 readers are out of scope: SURVEY.md section 8f rank 2).
"""
from __future__ import annotations

import time
from pathlib import Path

import numpy as np

from . import fem, ufl
from .problem import NonlinearProblem


def phi_set(x):
    """Obstacle of obstacle_pg.py:92-104."""
    r = np.sqrt(x[0] ** 2 + x[1] ** 2)
    r0 = 0.5
    beta = 0.9
    b = r0 * beta
    tmp = np.sqrt(r0**2 - b**2)
    B = tmp + b * b / tmp
    C = -b / tmp
    cond_true = B + r * C
    with np.errstate(invalid="ignore"):
        cond_false = np.sqrt(r0**2 - r**2)
    true_indices = np.flatnonzero(r > b)
    cond_false[true_indices] = cond_true[true_indices]
    return cond_false


COLUMNS = ["Energy", "Complementarity", "Feasibility", "Dual Feasibility", "Newton steps", "Step sizes",
           "Primal increments", "Latent increments"]


def setup_problem(msh: fem.Mesh, polynomial_order: int = 1, petsc_options: dict | None = None, device: int = 0,
                  phi=phi_set, lu_comm=None, quadrature_degree: int = 6, quadrature_scheme: str | None = None):
    """Everything obstacle_pg.py does before the loop (:66-152). Returns (problem, sol, sol_k, alpha).
    quadrature_degree: 6 is the reference's (:106); 1 selects the vertex rule (mass lumping), with which the system is the
    finite-difference one of obstacle_finite_difference.jl (tests/test_gpu_fd_pin.py).  quadrature_scheme names a table of
    tables/quadrature.json explicitly ("tri_deg6_12_b": the second admissible 12-point degree-6 rule)."""
    V = fem.functionspace(msh, ("Lagrange", polynomial_order), ncomp=2)  # :68-70
    alpha = fem.Constant(msh, 1.0)  # :73
    f = fem.Constant(msh, 0.0)  # :74
    dofs = msh.exterior_dofs(polynomial_order)  # :76-79
    bcs = fem.dirichletbc(0.0, dofs, V.sub(0))  # :81-83
    sol, sol_k = fem.Function(V), fem.Function(V)  # :86-87
    phi_fn = fem.QuadratureFunction(msh, quadrature_degree, name="phi", scheme=quadrature_scheme)
    phi_fn.interpolate(phi)  # :110-111
    # the residual as the reference states it (:88-89,114-125), through the UFL-subset front end (ufl.py), which selects the
    # HIP kernel family for it
    u, psi = ufl.split(sol)
    u_k, psi_k = ufl.split(sol_k)
    v, w = ufl.TestFunctions(V)
    dx = ufl.Measure("dx", domain=msh, metadata={"quadrature_degree": quadrature_degree})
    F = (alpha * ufl.inner(ufl.grad(u), ufl.grad(v)) * dx + psi * v * dx + u * w * dx - ufl.exp(psi) * w * dx
         - phi_fn * w * dx - alpha * f * v * dx - psi_k * v * dx)
    J = ufl.derivative(F, sol)
    if petsc_options is None:
        petsc_options = {  # :128-139
            "ksp_type": "preonly",
            "pc_type": "lu",
            "ksp_error_if_not_converged": True,
            "snes_error_if_not_converged": True,
            "snes_linesearch_type": "none",
            "snes_rtol": 1e-6,
            "snes_max_it": 100,
        }
    problem = NonlinearProblem(F, u=sol, bcs=[bcs], J=J, petsc_options=petsc_options,
                               petsc_options_prefix="obstacle_", device=device, lu_comm=lu_comm)  # :140-142
    return problem, sol, sol_k, alpha


def alpha_update(rule, k, alpha_k, alpha_max, C=1.0, r=1.5, q=1.5, current=1.0):
    """Step-size rules of obstacle_pg.py:173-186. Returns (alpha.value, alpha_k)."""
    if rule == "constant":
        return C, alpha_k
    if rule == "double_exponential":
        value = current
        try:
            value = max(C * r ** (q**k) - alpha_k, C)
        except OverflowError:
            pass
        alpha_k = value
        return min(value, alpha_max), alpha_k
    return C * r**k, alpha_k  # "geometric" (the reference's else branch)


def run_outer_loop(problem, sol, sol_k, alpha, maximum_number_of_outer_loop_iterations, alpha_scheme, alpha_max,
                   tol_exit, device_resident=True, verbose=False):
    """obstacle_pg.py:154-227. With device_resident=False the iterate update is the reference's literal
    `sol_k.x.array[:] = sol.x.array[:]` (a PCIe round trip); True does the same copy on the device."""
    hist = {c: [] for c in COLUMNS}
    if device_resident:
        problem.zero_state()  # :157-158 on the device
    else:
        sol.x.array[:] = 0.0  # :157
        sol_k.x.array[:] = sol.x.array[:]  # :158
    alpha_k = 1
    increment_k = 0.0
    k = -1
    for k in range(maximum_number_of_outer_loop_iterations):
        alpha.value, alpha_k = alpha_update(alpha_scheme, k, alpha_k, alpha_max, current=alpha.value)
        if verbose:
            print(f"OUTER LOOP {k + 1} alpha: {alpha.value}")
        problem.solve()  # :190
        converged_reason = problem.solver.getConvergedReason()
        n = problem.solver.getIterationNumber()
        if verbose:
            print(f"Newton steps: {n}   Converged: {converged_reason}")
        energy, complementarity, feasibility, dual_feasibility, increment, latent_increment = problem.observables()
        if verbose:
            msg = f"Increment size: {increment}"
            if increment_k > 0.0:
                msg += f"   Ratio: {increment / increment_k}"
            print(msg + "\n")
        for name, v in zip(COLUMNS, (energy, complementarity, feasibility, dual_feasibility, n, float(alpha.value),
                                     increment, latent_increment)):
            hist[name].append(v)
        if increment < tol_exit:  # :222-223
            break
        if device_resident:
            sol_k.x.assign_from(sol.x)  # :226 on the device
        else:
            sol_k.x.array[:] = sol.x.array[:]  # :226 literally
        increment_k = increment
    hist["outer_iterations"] = k + 1
    return hist


def solve_problem(msh, polynomial_order, maximum_number_of_outer_loop_iterations, alpha_scheme, alpha_max, tol_exit,
                  output_dir: Path | None = None, device_resident=True, verbose=True, return_history=False, device=0):
    """Reference signature (obstacle_pg.py:53-60) with the XDMF file name replaced by a Mesh."""
    problem, sol, sol_k, alpha = setup_problem(msh, polynomial_order, device=device)
    t0 = time.perf_counter()
    hist = run_outer_loop(problem, sol, sol_k, alpha, maximum_number_of_outer_loop_iterations, alpha_scheme,
                          alpha_max, tol_exit, device_resident, verbose)
    hist["loop_seconds"] = time.perf_counter() - t0
    nk = hist["outer_iterations"]
    num_primal_dofs = sol.function_space.block_size
    if output_dir is not None:  # CSV fingerprint, columns as obstacle_pg.py:244-258
        import pandas as pd

        output_dir = Path(output_dir)
        output_dir.mkdir(exist_ok=True, parents=True)
        df = pd.DataFrame({c: hist[c] for c in COLUMNS})
        df["Polynomial order"] = np.full(nk, polynomial_order)
        df["dofs"] = np.full(nk, num_primal_dofs)
        df["Step size rule"] = [alpha_scheme] * nk
        filename = output_dir / f"example_polyorder{polynomial_order}_{num_primal_dofs}.csv"
        if verbose:
            print(f"Saving data to: {filename}")
        df.to_csv(filename, index=False)
    if verbose and nk == maximum_number_of_outer_loop_iterations:
        print("Maximum number of outer loop iterations reached")
    sol.x.array  # make the host copy current before the backend goes away
    total = int(sum(hist["Newton steps"]))
    problem.close()
    if return_history:
        return sol, total, hist
    return sol, total
