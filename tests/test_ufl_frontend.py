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
        ufl.Measure("dS")  # interior-facet integrals: not in the subset
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


# ---- vector-valued latent variables (example 06) and composition (example 08) --------------------------------------------------
def _gc_setting(N=4):
    m = fem.create_unit_square(N, N)
    V = fem.functionspace(m, fem.mixed_element([fem.element("Lagrange", "triangle", 2), fem.element("Lagrange", "triangle", 1, shape=(2,))]))
    sol, w0 = fem.Function(V), fem.Function(V)
    U, U_to_W = V.sub(0).collapse()
    phi, f = fem.Function(U), fem.Function(U)
    return m, V, sol, w0, U, phi, f


def test_example06_form_is_recognised_however_it_is_written():
    """gradient_constraint_dolfinx.py:100-107 verbatim, and with the nonlinear term written as example 08 writes it."""
    m, V, sol, w0, U, phi, f = _gc_setting()
    assert V.num_dofs == 81 + 2 * 25 and V.component_rank(1) == 1 and U.num_dofs == 81
    u, psi = ufl.split(sol)
    v, w = ufl.TestFunctions(V)
    _, psi0 = ufl.split(w0)
    dx = ufl.Measure("dx", domain=m, metadata={"quadrature_degree": 10})
    alpha = fem.Constant(m, 1.0)
    F = alpha * ufl.inner(ufl.grad(u), ufl.grad(v)) * dx
    F += ufl.inner(psi, ufl.grad(v)) * dx
    F -= alpha * ufl.inner(f, v) * dx
    F -= ufl.inner(psi0, ufl.grad(v)) * dx
    F += ufl.inner(ufl.grad(u), w) * dx
    non_lin_term = 1 / (ufl.sqrt(1 + ufl.dot(psi, psi)))
    F -= phi * non_lin_term * ufl.dot(psi, w) * dx
    spec = ufl.compile_form(F, sol)
    assert isinstance(spec, ufl.GradientConstraintSpec)
    assert spec.phi is phi and spec.f is f and spec.w0 is w0 and spec.alpha is alpha and spec.quadrature_degree == 10
    G = (alpha * ufl.inner(ufl.grad(u), ufl.grad(v)) * dx - alpha * ufl.inner(f, v) * dx
         + ufl.compose(ufl.Form([]), [("hellinger", u, psi, psi0, v, w, phi, dx)]))
    assert isinstance(ufl.compile_form(G, sol), ufl.GradientConstraintSpec) and ufl.forms_equal(F, G)
    # phi and f swapped, a sign flipped, exp in place of the Hellinger map: none of these is example 06
    bad1 = F + alpha * ufl.inner(f, v) * dx - alpha * ufl.inner(phi, v) * dx + phi * non_lin_term * ufl.dot(psi, w) * dx - f * non_lin_term * ufl.dot(psi, w) * dx
    assert ufl.compile_form(bad1, sol).phi is f  # consistently swapped roles are simply the other assignment
    for bad in (F + 2.0 * ufl.inner(psi0, ufl.grad(v)) * dx, F + ufl.inner(psi, w) * dx):
        with pytest.raises(NotImplementedError):
            ufl.compile_form(bad, sol)


def test_example08_residual_is_the_composition_of_the_obstacle_and_the_gradient_rows():
    """SURVEY's acceptance test for the front end: intersecting_constraints_dolfinx.py:17-58 - three components (u, psi0, psi),
    energy derivative, BOTH latent maps - stated verbatim (on a 2-D mesh: the Mesh type here is 2-D; the form is dimension-blind)
    equals, as a polynomial in its terminals, the primal rows + example 01's latent rows for psi0 + example 06's for psi."""
    mesh = fem.create_unit_square(3, 3)
    el_s = fem.element("Lagrange", "triangle", 1)
    el_v = fem.element("Lagrange", "triangle", 1, shape=(2,))
    Z = fem.functionspace(mesh, fem.mixed_element([el_s, el_s, el_v]))
    z = fem.Function(Z, name="Solution")
    (u, psi0, psi) = ufl.split(z)
    z_test = ufl.TestFunction(Z)
    (v, w0, w) = ufl.split(z_test)
    z_iter = fem.Function(Z, name="PreviousLVPPSolution")
    (u_iter, psi0_iter, psi_iter) = ufl.split(z_iter)
    c = fem.Constant(mesh, 0.0)
    dx = ufl.dx(domain=mesh)
    E = 0.5 * ufl.inner(ufl.grad(u), ufl.grad(u)) * dx + c * u * dx
    x = ufl.SpatialCoordinate(mesh)[0]
    (l, r) = (0.2, 0.8)
    bump = ufl.exp(-1 / (10 * (x - l) * (r - x))) / ufl.exp(-1 / (10 * (0.5 - l) * (r - 0.5)))
    phi0 = ufl.conditional(ufl.le(x, l), 0, ufl.conditional(ufl.ge(x, r), 0, bump))
    phic = fem.Constant(mesh, 100.0)
    phi = ufl.conditional(ufl.le(x, 0.2), phic, ufl.conditional(ufl.gt(x, 0.8), phic, 100))
    alpha = fem.Constant(mesh, 1.0)
    F = (
        alpha * ufl.derivative(E, z, z_test)
        + ufl.inner(psi0, v) * dx
        + ufl.inner(psi, ufl.grad(v)) * dx
        - ufl.inner(psi0_iter, v) * dx
        - ufl.inner(psi_iter, ufl.grad(v)) * dx
        + ufl.inner(u, w0) * dx
        - ufl.inner(ufl.exp(psi0), w0) * dx
        - ufl.inner(phi0, w0) * dx
        + ufl.inner(ufl.grad(u), w) * dx
        - ufl.inner(phi * psi / ufl.sqrt(1 + ufl.dot(psi, psi)), w) * dx
    )
    primal = alpha * ufl.inner(ufl.grad(u), ufl.grad(v)) * dx + alpha * c * v * dx  # = alpha dE/du[v]
    G = ufl.compose(primal, [("exp", u, psi0, psi0_iter, v, w0, phi0, dx), ("hellinger", u, psi, psi_iter, v, w, phi, dx)])
    assert ufl.forms_equal(F, G)
    # and it is NOT the composition with the maps exchanged or with one constraint dropped
    assert not ufl.forms_equal(F, ufl.compose(primal, [("exp", u, psi0, psi0_iter, v, w0, phi0, dx)]))
    H = ufl.compose(primal, [("exp", u, psi0, psi0_iter, v, w0, phi, dx), ("hellinger", u, psi, psi_iter, v, w, phi0, dx)])
    assert not ufl.forms_equal(F, H)
    # selection recognises the family on any mesh (the form is dimension-blind): phi0 and phi come back as data expressions
    spec = ufl.compile_form(F, z)
    assert isinstance(spec, ufl.IntersectingSpec) and spec.alpha is alpha and spec.c is c and spec.z_iter is z_iter
    pts = np.array([[0.1, 0.5, 0.9], [0.3, 0.3, 0.3]])
    assert np.allclose(spec.phi(pts), [100.0, 100.0, 100.0]) and np.allclose(spec.phi0(pts), [0.0, 1.0, 0.0])
    phic.value = 0.5
    assert np.allclose(spec.phi(pts), [0.5, 100.0, 0.5])  # the Constant inside the expression stays live
    # ... but its HIP kernels (include/pgx_ic.h) are written for the reference's mesh, an interval
    from proximalgalerkin_amd.intersecting import NonlinearProblem as ICProblem

    with pytest.raises(NotImplementedError, match="interval"):
        ICProblem(F, z, bcs=[fem.dirichletbc(0.0, mesh.exterior_dofs(1), Z.sub(0))])


def test_example08_script_statement_compiles_on_the_interval():
    """intersecting_constraints_dolfinx.py:13-63 verbatim (proximalgalerkin_amd.intersecting.build_forms): the front end lifts the
    two coordinate expressions, matches the rest against compose(primal, [exp rows, Hellinger rows]), and refuses near misses."""
    from proximalgalerkin_amd import intersecting as I

    P = I.build_forms(20)
    spec = ufl.compile_form(P["F"], P["z"])
    assert isinstance(spec, ufl.IntersectingSpec) and spec.alpha is P["alpha"] and spec.z_iter is P["z_iter"]
    assert spec.quadrature_degree is None  # left to the estimate: degree 6 (oracle/ic_oracle.py)
    x = np.linspace(0, 1, 41)[None]
    assert np.array_equal(spec.phi0(x), I.phi0_bump(x)) and np.array_equal(spec.phi(x), I.phi_bound(100.0)(x))
    z, Z = P["z"], P["Z"]
    (u, psi0, psi), (v, w0, w) = ufl.split(z), ufl.split(ufl.TestFunction(Z))
    dx = ufl.dx(domain=P["mesh"])
    for bad in (P["F"] + ufl.inner(psi0, w0) * dx,                      # an extra mass term in the latent row
                P["F"] - 2.0 * ufl.inner(ufl.exp(psi0), w0) * dx,      # wrong weight of the exp map
                P["F"] + ufl.inner(psi, ufl.grad(v)) * dx):            # the gradient coupling doubled
        with pytest.raises(NotImplementedError):
            ufl.compile_form(bad, z)
    # the obstacle may be ANY expression of the coordinate: it is data, not part of the family
    xs = ufl.SpatialCoordinate(P["mesh"])[0]
    (_, psi0_iter, psi_iter) = ufl.split(P["z_iter"])
    other = ufl.compose(P["alpha"] * ufl.inner(ufl.grad(u), ufl.grad(v)) * dx,
                        [("exp", u, psi0, psi0_iter, v, w0, ufl.sin(ufl.pi * xs), dx), ("hellinger", u, psi, psi_iter, v, w, 2 + xs * xs, dx)])
    spec2 = ufl.compile_form(other, z)
    assert spec2.c is None and np.allclose(spec2.phi0(x), np.sin(np.pi * x[0])) and np.allclose(spec2.phi(x), 2 + x[0] ** 2)


def test_gateaux_derivative_of_an_energy():
    m, V, sol, w0, U, phi, f = _gc_setting()
    u, psi = ufl.split(sol)
    zt = ufl.TestFunction(V)
    v, w = ufl.split(zt)
    dx = ufl.dx(domain=m)
    E = 0.5 * ufl.inner(ufl.grad(u), ufl.grad(u)) * dx + 0.5 * ufl.inner(psi, psi) * dx + ufl.exp(u) * dx - f * u * dx
    dE = ufl.derivative(E, sol, zt)
    expect = ufl.inner(ufl.grad(u), ufl.grad(v)) * dx + ufl.inner(psi, w) * dx + ufl.exp(u) * v * dx - f * v * dx
    assert ufl.forms_equal(dE, expect)


# ---- example 02: tensor algebra, facet measure, blocked spaces -----------------------------------------------------------------
def _signorini_forms(E=2.0e4, nu=0.3, gap=0.1, alpha_0=1.0, swap=False):
    """signorini_dolfinx.py:146-153 and :199-252 verbatim (native 3-D branch)."""
    from proximalgalerkin_amd import signorini as sg

    mesh = sg.create_unit_cube(3, 2, 2)
    facet_tag, boundary_conditions = sg.native_tags(mesh)

    def epsilon(w):
        return ufl.sym(ufl.grad(w))

    def sigma(w, mu, lmbda, gdim):
        return 2.0 * mu * epsilon(w) + lmbda * ufl.tr(ufl.grad(w)) * ufl.Identity(gdim)

    contact_facets = np.concatenate([facet_tag.find(m) for m in boundary_conditions["contact"]])
    gdim = 3
    submesh, submesh_to_mesh = fem.create_submesh(mesh, 2, contact_facets)
    ds = ufl.Measure("ds", domain=mesh, subdomain_data=facet_tag, subdomain_id=boundary_conditions["contact"],
                     metadata={"quadrature_degree": 4})
    V = fem.functionspace(mesh, ("Lagrange", 1, (gdim,)))
    W = fem.functionspace(submesh, ("Lagrange", 1))
    Q = ufl.MixedFunctionSpace(V, W)
    v, w = ufl.TestFunctions(Q)
    u, psi, psi_k = fem.Function(V, name="displacement"), fem.Function(W), fem.Function(W)
    mu = E / (2.0 * (1.0 + nu))
    lmbda = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
    n_g = fem.Constant(mesh, np.zeros(gdim))
    n_g.value[-1] = -1
    alpha = fem.Constant(mesh, alpha_0)
    f = fem.Constant(mesh, np.zeros(gdim))
    x = ufl.SpatialCoordinate(mesh)
    g = x[gdim - 1] + fem.Constant(mesh, -gap)
    residual = alpha * ufl.inner(sigma(u, mu, lmbda, gdim), epsilon(v)) * ufl.dx(domain=mesh) - alpha * ufl.inner(f, v) * ufl.dx(domain=mesh)
    residual += -ufl.inner(psi - psi_k, ufl.dot(v, n_g)) * ds
    residual += ufl.inner(ufl.dot(u, n_g), w) * ds
    residual += ufl.inner(ufl.exp(psi), w) * ds - ufl.inner(g, w) * ds
    if swap:
        residual += 2.0 * ufl.inner(ufl.exp(psi), w) * ds
    return ufl.extract_blocks(residual), u, psi, psi_k, alpha, mu, lmbda, contact_facets, mesh, facet_tag, boundary_conditions


def test_example02_blocked_residual_is_recognised_and_lame_constants_are_read_off():
    F, u, psi, psi_k, alpha, mu, lmbda, contact, *_ = _signorini_forms()
    spec = ufl.compile_signorini(F, [u, psi])
    assert spec.u is u and spec.psi is psi and spec.psi_k is psi_k and spec.alpha is alpha
    assert spec.mu == pytest.approx(mu, rel=1e-14) and spec.lmbda == pytest.approx(lmbda, rel=1e-14)
    assert spec.gap == pytest.approx(0.1) and spec.quadrature_degree == 4
    assert np.array_equal(spec.contact_facets, contact)
    assert u.function_space.num_dofs == 3 * 36 and psi.function_space.num_dofs == 12
    with pytest.raises(NotImplementedError):  # e^psi with the wrong sign / weight is not the Signorini residual
        ufl.compile_signorini(_signorini_forms(swap=True)[0], [u, psi])


def test_tensor_identities_of_the_canonical_form():
    """inner(lambda tr(grad u) I, sym(grad v)) = lambda div(u) div(v); inner(sym(grad u), sym(grad v)) is symmetric in (u, v)."""
    F, u, psi, *_ = _signorini_forms()
    V = u.function_space
    Q = ufl.MixedFunctionSpace(V, psi.function_space)
    v, w = ufl.TestFunctions(Q)
    dx = ufl.dx(domain=V.mesh)
    a = ufl.inner(ufl.tr(ufl.grad(u)) * ufl.Identity(3), ufl.sym(ufl.grad(v))) * dx
    b = ufl.tr(ufl.grad(u)) * ufl.tr(ufl.sym(ufl.grad(v))) * dx
    assert ufl.forms_equal(a, b)
    c = ufl.inner(ufl.sym(ufl.grad(u)), ufl.sym(ufl.grad(v))) * dx
    d = ufl.inner(ufl.sym(ufl.grad(v)), ufl.sym(ufl.grad(u))) * dx
    assert ufl.forms_equal(c, d) and not ufl.forms_equal(a, c)
