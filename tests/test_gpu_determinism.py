"""Every assembly kernel is atomic-free: element contributions are parked in a stash and summed per destination in a fixed
order (csrc/pgx_scatter.h; P1 uses row-parallel owner-computes).  So residuals and Jacobians are BITWISE reproducible - run to
run and between two handles on the same input - which is what lets the replicas of a distributed-LU handle assemble redundantly
without exchanging anything (and what DOLFINx's sequential per-rank assembly gives the reference for free)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _twice(make, probe):
    out = []
    for _ in range(2):
        h = make()
        out.append([probe(h) for _ in range(2)])
        h.close()
    (a0, a1), (b0, b1) = out
    for x, y in ((a0, a1), (a0, b0), (a0, b1)):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)


def test_obstacle_p1_and_p2(require_gpu):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import setup_problem

    for degree, N in ((1, 96), (2, 64)):
        msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
        rng = np.random.default_rng(3)
        state = {}

        def make():
            problem, sol, sol_k, alpha = setup_problem(msh, degree)
            n2 = sol.function_space.num_dofs
            if "x" not in state:
                state["x"] = rng.standard_normal(n2) * 0.2
                state["x"][n2 // 2:] -= 5.0 * np.abs(rng.standard_normal(n2 // 2))
                state["xk"] = rng.standard_normal(n2) * 0.2
            sol_k.x.array[:] = state["xk"]
            alpha.value = 3.5
            return problem

        def probe(problem):
            F, _ = problem.residual(state["x"])
            problem.assemble_jacobian(state["x"])
            return [F, problem.export_blocks()[4], problem.spmv(state["xk"])]

        _twice(make, probe)


def test_gradient_constraint(require_gpu):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default

    mesh = fem.create_unit_square(40, 40)
    rng = np.random.default_rng(4)
    st = {}

    def make():
        p = GradientConstraintProblem(mesh, phi_default, f_default)
        if "x" not in st:
            st["x"], st["xk"] = rng.standard_normal(p.ndofs) * 2.0, rng.standard_normal(p.ndofs)
        p.set_alpha(8.0)
        p.set_prev(st["xk"])
        return p

    _twice(make, lambda p: [p.residual(st["x"])[0], p.jacobian(st["x"]).data])


def test_signorini(require_gpu):
    from proximalgalerkin_amd import signorini as G

    mesh = G.create_unit_cube(7, 6, 5)
    mt, _ = G.native_tags(mesh)
    rng = np.random.default_rng(5)
    st = {}

    def make():
        p = G.SignoriniProblem(mesh, mt.find(2), np.unique(mt.find(1).ravel()), 2.0e4, 0.3, 0.01, -0.25)
        if "x" not in st:
            st["x"], st["xk"] = rng.standard_normal(p.ndofs) * 0.05, rng.standard_normal(p.ndofs) * 0.05
            st["x"][3 * mesh.geometry.shape[0]:] -= 3.0
        p.set_alpha(4.0)
        p.set_prev(st["xk"])
        return p

    _twice(make, lambda p: [p.residual(st["x"])[0], p.jacobian(st["x"]).data])


def test_thermoforming(require_gpu):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.thermoforming import ThermoformingProblem

    mesh = fem.create_unit_square(30, 30)
    rng = np.random.default_rng(6)
    st = {}

    def make():
        p = ThermoformingProblem(mesh)
        if "x" not in st:
            st["x"], st["xk"] = rng.standard_normal(p.ndofs) * 0.3, rng.standard_normal(p.ndofs) * 0.3
        p.set_alpha(2.0)
        p.set_prev(st["xk"])
        return p

    _twice(make, lambda p: [p.residual(st["x"])[0], p.jacobian(st["x"]).data])
