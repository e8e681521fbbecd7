"""UFL-subset front end (proximalgalerkin_amd/ufl.py): forms written as in the reference
(/root/reference/examples/01_obstacle_problem/obstacle_pg.py:88-125) are canonicalised and matched to the HIP kernel
family; anything else is refused with the offending terms named.  Host logic only - no GPU."""
import numpy as np
import pytest

from proximalgalerkin_amd import fem, ufl
from proximalgalerkin_amd.problem import ObstacleResidual


def _setting(degree=6):
    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (4, 4))
    V = fem.functionspace(msh, ("Lagrange", 1), ncomp=2)
    sol, sol_k = fem.Function(V), fem.Function(V)
    alpha, f = fem.Constant(msh, 1.0), fem.Constant(msh, 0.0)
    phi = fem.QuadratureFunction(msh, degree)
    dx = ufl.Measure("dx", domain=msh, metadata={"quadrature_degree": degree})
    return msh, V, sol, sol_k, alpha, f, phi, dx


def test_reference_statement_of_the_obstacle_form_is_recognised():
    msh, V, sol, sol_k, alpha, f, phi, dx = _setting()
    u, psi = ufl.split(sol)
    u_k, psi_k = ufl.split(sol_k)
    v, w = ufl.TestFunctions(V)
    F = (alpha * ufl.inner(ufl.grad(u), ufl.grad(v)) * dx + psi * v * dx + u * w * dx - ufl.exp(psi) * w * dx - phi * w * dx
         - alpha * f * v * dx - psi_k * v * dx)
    spec = ufl.compile_form(F, sol)
    assert isinstance(spec, ObstacleResidual)
    assert spec.sol is sol and spec.sol_k is sol_k and spec.alpha is alpha and spec.f is f and spec.phi is phi
    assert spec.quadrature_degree == 6


def test_equal_forms_written_differently_are_recognised():
    """Regrouped (thermoforming_dolfinx.py:57-62 style: inner() of scalars, += of partial sums, factored terms)."""
    msh, V, sol, sol_k, alpha, f, phi, dx = _setting()
    u, psi = ufl.split(sol)
    _, psi_k = ufl.split(sol_k)
    v, w = ufl.TestFunctions(V)
    F = ufl.inner(alpha * ufl.grad(u), ufl.grad(v)) * dx + ufl.inner(psi - psi_k, v) * dx
    F += -alpha * ufl.inner(f, v) * dx
    F += (u - ufl.exp(psi) - phi) * w * dx
    spec = ufl.compile_form(F, sol)
    assert spec.alpha is alpha and spec.f is f
    # alpha * (form) and 2*x - x
    G = alpha * (ufl.inner(ufl.grad(u), ufl.grad(v)) * dx - f * v * dx) + (2 * psi * v - psi * v - psi_k * v) * dx \
        + ufl.inner(w, u - phi) * dx - w * ufl.exp(psi) * dx
    spec = ufl.compile_form(G, sol)
    assert spec.alpha is alpha and spec.f is f and spec.phi is phi


def test_source_term_may_be_left_out():
    msh, V, sol, sol_k, alpha, f, phi, dx = _setting()
    u, psi = ufl.split(sol)
    _, psi_k = ufl.split(sol_k)
    v, w = ufl.TestFunctions(V)
    F = alpha * ufl.inner(ufl.grad(u), ufl.grad(v)) * dx + (psi - psi_k) * v * dx + (u - ufl.exp(psi) - phi) * w * dx
    spec = ufl.compile_form(F, sol)
    assert spec.alpha is alpha and spec.f.value == 0.0


@pytest.mark.parametrize("mutation", ["sign", "coefficient", "wrong_map", "extra_term", "swapped_tests", "unscaled_source"])
def test_other_forms_are_refused_with_the_terms_named(mutation):
    msh, V, sol, sol_k, alpha, f, phi, dx = _setting()
    u, psi = ufl.split(sol)
    _, psi_k = ufl.split(sol_k)
    v, w = ufl.TestFunctions(V)
    stiff = alpha * ufl.inner(ufl.grad(u), ufl.grad(v)) * dx
    prox = (psi - psi_k) * v * dx
    src = -alpha * f * v * dx
    lat = (u - ufl.exp(psi) - phi) * w * dx
    if mutation == "sign":
        F = stiff + prox + src + (u + ufl.exp(psi) - phi) * w * dx
    elif mutation == "coefficient":
        F = stiff + 2.0 * prox + src + lat
    elif mutation == "wrong_map":
        F = stiff + prox + src + (u - ufl.exp(-psi) - phi) * w * dx
    elif mutation == "extra_term":
        F = stiff + prox + src + lat + u * v * dx
    elif mutation == "swapped_tests":
        F = alpha * ufl.inner(ufl.grad(u), ufl.grad(w)) * dx + prox + src + lat
    else:
        F = stiff + prox - f * v * dx + lat
    with pytest.raises(NotImplementedError) as e:
        ufl.compile_form(F, sol)
    assert "obstacle (example 01)" in str(e.value) and "quadrature degree 6" in str(e.value)


def test_measure_and_structure_errors():
    msh, V, sol, sol_k, alpha, f, phi, dx = _setting()
    u, psi = ufl.split(sol)
    v, w = ufl.TestFunctions(V)
    with pytest.raises(NotImplementedError):
        ufl.Measure("ds")
    with pytest.raises(ValueError):
        ufl.grad(u) * ufl.grad(v)
    with pytest.raises(ValueError):
        u + ufl.grad(u)
    with pytest.raises(NotImplementedError):  # no explicit quadrature degree
        ufl.compile_form(u * v * ufl.Measure("dx"), sol)
    with pytest.raises(TypeError):
        sol_k * dx  # a mixed Function enters through split()


def test_canonical_form_is_a_polynomial_identity():
    msh, V, sol, sol_k, alpha, f, phi, dx = _setting()
    u, psi = ufl.split(sol)
    v, w = ufl.TestFunctions(V)
    a = (u + psi) * (v - w) * dx
    b = u * v * dx - u * w * dx + psi * v * dx - w * psi * dx
    names = {t.key(): (n, False) for t, n in zip(ufl.terminals(a), "abcd")}
    assert ufl.canonical(a, names) == ufl.canonical(b, names)
    c = ufl.inner(ufl.grad(alpha * u / 2.0), ufl.grad(v)) * dx
    d = 0.5 * alpha * ufl.inner(ufl.grad(v), ufl.grad(u)) * dx
    names = {t.key(): (n, t.kind == "constant") for t, n in zip(ufl.terminals(c), "pqr")}
    assert ufl.canonical(c, names) == ufl.canonical(d, names)
    assert float(alpha) == 1.0 and np.isfinite(list(ufl.canonical(c, names).values())).all()


def _thermoforming_forms(eps_in_residual=False, wrong=False):
    from proximalgalerkin_amd.ufl import conditional, dx, exp, grad, inner, lt, max_value, pi, sin

    mesh = fem.create_unit_square(3, 3)
    V = fem.functionspace(mesh, ("Lagrange", 1), ncomp=3)
    s, s_prev = fem.Function(V), fem.Function(V)
    u, T, psi = ufl.split(s)
    v, q, w = ufl.TestFunctions(V)
    _, _, psi_prev = ufl.split(s_prev)
    c = {k: fem.Constant(mesh, val) for k, val in dict(bound0=0.0, bound1=0.01, beta=1.0, alpha=2.0**-6, f=25.0, eps=1e-10).items()}

    def g(t):
        return conditional(lt(t, c["bound0"]), 1, conditional(lt(t, c["bound1"]), 1 - t / c["bound1"], 0))

    x, y = ufl.SpatialCoordinate(mesh)
    Phi0 = 1 - 2 * max_value(abs(x - 0.5), abs(y - 0.5))
    xi = sin(pi * x) * sin(pi * (x if wrong else y))
    F = c["alpha"] * inner(grad(u), grad(v)) * dx + inner(psi, v) * dx
    F += -c["alpha"] * inner(c["f"], v) * dx - inner(psi_prev, v) * dx
    F += inner(grad(T), grad(q)) * dx + c["beta"] * inner(T, q) * dx
    F += -inner(g(exp(-psi)), q) * dx
    F += inner(u, w) * dx + inner(exp(-psi), w) * dx
    F += -inner(Phi0 + xi * T, w) * dx
    G = F - c["eps"] / c["alpha"] * inner(grad(psi), grad(w)) * dx
    return (G if eps_in_residual else F), G, s, s_prev, c


def test_thermoforming_forms_and_modified_jacobian_are_recognised():
    """thermoforming_dolfinx.py:36-71: five Constants are told apart by where they stand in the form; the Jacobian form
    differs from the residual by -eps/alpha*inner(grad(psi), grad(w))."""
    F, G, s, s_prev, c = _thermoforming_forms()
    spec = ufl.compile_form(F, s, ufl.derivative(G, s))
    assert isinstance(spec, ufl.ThermoformingSpec)
    for k in ("alpha", "beta", "f", "bound0", "bound1", "eps"):
        assert getattr(spec, k) is c[k], k
    assert spec.s is s and spec.s_prev is s_prev and spec.quadrature_degree is None
    assert ufl.compile_form(F, s, ufl.derivative(F, s)).eps is None  # exact Jacobian
    assert ufl.compile_form(F, s).eps is None


def test_thermoforming_variants_are_refused():
    F, G, s, s_prev, c = _thermoforming_forms(eps_in_residual=True)
    with pytest.raises(NotImplementedError, match="belongs in the Jacobian"):
        ufl.compile_form(F, s)
    F, G, s, s_prev, c = _thermoforming_forms(wrong=True)  # xi = sin(pi x) sin(pi x)
    with pytest.raises(NotImplementedError, match="thermoforming QVI"):
        ufl.compile_form(F, s)
    F, G, s, s_prev, c = _thermoforming_forms()
    other = fem.Function(s.function_space)
    with pytest.raises(NotImplementedError):
        ufl.compile_form(F, s, ufl.derivative(G, other))
