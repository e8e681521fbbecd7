"""HIP path vs the reference's own finite-difference statement of LVPP (obstacle_finite_difference.jl, transcribed in
oracle/fd_oracle.py).  With the vertex quadrature rule the P1 system the HIP kernels assemble IS that finite-difference
system, row-scaled (tests/test_oracle_fd.py proves it for the oracle), with alpha_fe = 4 alpha_fd.  The Julia script's loop
(alpha rule, 1e-4 relative-residual Newton test, 1e-9 stopping test, psi_0 = 1) is driven from fd_oracle; each Newton step is
ONE pgx_newton_solve step on the GPU (assembly, Jacobian, multigrid-preconditioned FGMRES, update).  Bar: identical Newton
counts per proximal step, every iterate and the final u within 1e-10 relative l2 of the transcription's.

N = 257 (66 049 vertices) is above `fused_min`, so the fused multi-sweep smoother and residual+restriction kernels that
dominate the 2048^2 benchmark are the ones compared here."""
import numpy as np
import pytest

from oracle import fd_oracle as FD

pytestmark = pytest.mark.gpu


def hip_newton_step(N, pc=None):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import setup_problem

    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N - 1, N - 1))
    opts = {"snes_linesearch_type": "none", "snes_rtol": 1e30, "snes_max_it": 1, "ksp_rtol": 1e-13}
    if pc:
        opts["pc_type"] = pc
    problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=opts, phi=lambda x: FD.phi(x[0], x[1]),
                                               quadrature_degree=1)
    n = N * N

    def step(u, psi, alpha_fd, w):
        sol.x.array[:n] = u
        sol.x.array[n:] = psi
        sol_k.x.array[:n] = 0.0
        sol_k.x.array[n:] = w
        alpha.value = 4.0 * alpha_fd  # the script's stencil is 4 x the Laplacian of its own grid (fd_oracle.py header)
        problem.solve()
        assert problem.solver.getConvergedReason() == 3 and problem.solver.getIterationNumber() == 1
        x = sol.x.array
        return x[:n].copy(), x[n:].copy()

    return step, problem


@pytest.mark.parametrize("N,pc", [(17, None), (65, None), (65, "pgx_lu"), (257, None)])
def test_hip_path_reproduces_the_finite_difference_lvpp_run(require_gpu, N, pc):
    rec_fd, rec = [], []
    _, U, its, per = FD.fd_lvpp_solve(N, record=rec_fd)
    step, problem = hip_newton_step(N, pc)
    _, U2, its2, per2 = FD.fd_lvpp_solve(N, newton_step=step, record=rec)
    problem.close()
    assert per2 == per and its2 == its, (per, per2)
    worst = 0.0
    for (k, i, u, p), (k2, i2, u2, p2) in zip(rec_fd, rec):
        assert (k, i) == (k2, i2)
        worst = max(worst, np.linalg.norm(u - u2) / np.linalg.norm(u))
    assert worst < 1e-10, worst
    assert np.linalg.norm(U - U2) < 1e-10 * np.linalg.norm(U)
