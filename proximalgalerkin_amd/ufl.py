"""UFL-subset front end: symbolic residual forms in, kernel selection out.

The reference states its problems as UFL forms and lets FFCx generate element kernels
(/root/reference/examples/01_obstacle_problem/obstacle_pg.py:88-125).  Here the element kernels are hand-written
HIP, one set per problem family, so the front end does not generate code: it

  1. builds an expression tree for the operator subset the in-scope examples use (split, TestFunctions, inner, dot, grad,
     exp, sqrt, sin, abs, max_value, conditional, lt, SpatialCoordinate, arithmetic, Measure("dx", metadata=
     {"quadrature_degree": q}), derivative) - SURVEY.md App. B,
  2. expands a form into a canonical sum of monomials  coef * (scalar factors) * [inner(vector, vector)]  per measure
     (bilinearity of inner, linearity of grad, constants pulled out of grad, like terms combined), so that forms that
     are EQUAL as polynomials in their terminals compare equal however they were written, and
  3. matches the canonical form against the registered family templates - written in this same language - by unifying
     the user's terminals (unknown, previous iterate, Constants, quadrature-space coefficients) with the template's roles.

A match returns the declarative description the HIP path is built from (problem.ObstacleResidual for example 01,
ThermoformingSpec for example 05, whose Jacobian form differs from the derivative of the residual); anything else raises
NotImplementedError naming the terms that do not fit, never a silent approximation.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass

import numpy as np

from . import fem

# ---------------------------------------------------------------------------------------------------------------------
# expression tree
# ---------------------------------------------------------------------------------------------------------------------


class Expr:
    rank = 0  # 0 scalar, 1 vector

    def __add__(self, o):
        return Sum(self, as_expr(o))

    def __radd__(self, o):
        return Sum(as_expr(o), self)

    def __sub__(self, o):
        return Sum(self, Scaled(-1.0, as_expr(o)))

    def __rsub__(self, o):
        return Sum(as_expr(o), Scaled(-1.0, self))

    def __neg__(self):
        return Scaled(-1.0, self)

    def __mul__(self, o):
        if isinstance(o, Measure):
            return Form([Integral(self, o)])
        if isinstance(o, Form):
            return o.__rmul__(self)
        return Product(self, as_expr(o))

    def __rmul__(self, o):
        return Product(as_expr(o), self)

    def __truediv__(self, o):
        return Division(self, as_expr(o))

    def __rtruediv__(self, o):
        return Division(as_expr(o), self)


class Number(Expr):
    def __init__(self, v):
        self.v = float(v)


class Terminal(Expr):
    """kind: 'component' (obj = fem.Function, index), 'argument' (obj = FunctionSpace, index), 'constant' (fem.Constant),
    'quadrature' (fem.QuadratureFunction), 'role' (template placeholder, obj = role name)."""

    def __init__(self, kind, obj, index=0, rank=0):
        self.kind, self.obj, self.index, self.rank = kind, obj, index, rank

    def key(self):
        return (self.kind, id(self.obj) if self.kind != "role" else self.obj, self.index)


class Sum(Expr):
    def __init__(self, a, b):
        if a.rank != b.rank:
            raise ValueError("cannot add a scalar and a vector expression")
        self.a, self.b, self.rank = a, b, a.rank


class Scaled(Expr):
    def __init__(self, c, a):
        self.c, self.a, self.rank = float(c), a, a.rank


class Product(Expr):
    def __init__(self, a, b):
        if a.rank and b.rank:
            raise ValueError("product of two vectors: use inner() or dot()")
        self.a, self.b, self.rank = a, b, max(a.rank, b.rank)


class Division(Expr):
    def __init__(self, a, b):
        if b.rank:
            raise ValueError("division by a vector")
        self.a, self.b, self.rank = a, b, a.rank


class Grad(Expr):
    def __init__(self, a):
        if a.rank > 1:
            raise NotImplementedError("grad of a tensor expression")
        self.a, self.rank = a, a.rank + 1  # grad of a vector field is a rank-2 tensor (signorini_dolfinx.py:146-153)


class Sym(Expr):
    def __init__(self, a):
        if a.rank != 2:
            raise ValueError("sym() of a non-tensor")
        self.a, self.rank = a, 2


class Tr(Expr):
    def __init__(self, a):
        if a.rank != 2:
            raise ValueError("tr() of a non-tensor")
        self.a, self.rank = a, 0


class IdentityTensor(Expr):
    def __init__(self, dim):
        self.dim, self.rank = int(dim), 2


class Inner(Expr):
    def __init__(self, a, b):
        if a.rank != b.rank:
            raise ValueError("inner() of operands of different rank")
        self.a, self.b, self.rank = a, b, 0


class Func(Expr):
    """exp, sqrt, sin, abs, max_value, lt, conditional of scalar expressions: opaque to the polynomial expansion, compared
    through the canonical text of the arguments."""

    def __init__(self, name, *args):
        if any(a.rank for a in args):
            raise ValueError(f"{name} of a vector expression")
        self.name, self.args, self.rank = name, args, 0


def as_expr(o):
    if isinstance(o, Expr):
        return o
    if isinstance(o, (int, float)):
        return Number(o)
    if isinstance(o, fem.Constant):
        return Terminal("constant", o, 0, o.rank)
    if isinstance(o, fem.QuadratureFunction):
        return Terminal("quadrature", o)
    if isinstance(o, fem.Function):
        V = o.function_space
        if isinstance(V, fem.FunctionSpace) and V.ncomp == 1:  # a coefficient in a scalar space (phi, f of example 06)
            return Terminal("coefficient", o)
        if isinstance(V, (fem.VectorSpace, fem.FacetSpace)):  # blocked problems: each unknown is a Function of its own space
            return Terminal("field", o, 0, V.component_rank(0))
        raise TypeError("a mixed Function enters a form through split(function)")
    raise TypeError(f"cannot use {type(o).__name__} in a form")


def _comp_rank(V, i):
    return V.component_rank(i) if hasattr(V, "component_rank") else 0


def split(function: fem.Function):
    """ufl.split(sol) (obstacle_pg.py:88-89): the components of a mixed Function."""
    V = function.function_space
    return tuple(Terminal("component", function, i, _comp_rank(V, i)) for i in range(V.ncomp))


def TestFunctions(V: fem.FunctionSpace):
    """ufl.TestFunctions(V) (obstacle_pg.py:114)."""
    return tuple(Terminal("argument", V, i, _comp_rank(V, i)) for i in range(V.ncomp))


def grad(a):
    return Grad(as_expr(a))


def inner(a, b):
    return Inner(as_expr(a), as_expr(b))


def sym(a):
    return Sym(as_expr(a))


def tr(a):
    return Tr(as_expr(a))


def Identity(dim):
    return IdentityTensor(dim)


class MixedFunctionSpace:
    """ufl.MixedFunctionSpace(V, W) (signorini_dolfinx.py:225): a blocked space; TestFunctions(Q) gives one test function per block."""

    def __init__(self, *spaces):
        self.spaces = spaces
        self.ncomp = len(spaces)

    def component_rank(self, i):
        return self.spaces[i].component_rank(0)


def extract_blocks(form):
    """ufl.extract_blocks(residual) (signorini_dolfinx.py:252): the blocked residual; kept as one Form here (the families index the
    blocks by their test functions)."""
    return form


dot = inner  # real-valued: the same contraction for the ranks supported here


def exp(a):
    return Func("exp", as_expr(a))


def sqrt(a):
    return Func("sqrt", as_expr(a))


def sin(a):
    return Func("sin", as_expr(a))


def max_value(a, b):
    return Func("max_value", as_expr(a), as_expr(b))


def lt(a, b):
    return Func("lt", as_expr(a), as_expr(b))


def le(a, b):
    return Func("le", as_expr(a), as_expr(b))


def gt(a, b):
    return Func("gt", as_expr(a), as_expr(b))


def ge(a, b):
    return Func("ge", as_expr(a), as_expr(b))


def conditional(c, a, b):
    return Func("conditional", as_expr(c), as_expr(a), as_expr(b))


def _abs(a):
    return Func("abs", as_expr(a))


Expr.__abs__ = lambda self: _abs(self)
pi = 3.141592653589793


def SpatialCoordinate(mesh: fem.Mesh):
    """ufl.SpatialCoordinate(mesh) (thermoforming_dolfinx.py:51)."""
    return tuple(Terminal("coordinate", mesh, i) for i in range(mesh.geometry.shape[1]))


class Measure:
    """ufl.Measure("dx", domain=msh, metadata={"quadrature_degree": q}) (obstacle_pg.py:115); `ufl.dx` is the default cell
    measure (thermoforming_dolfinx.py:14), whose quadrature degree the family picks."""

    def __init__(self, name="dx", domain=None, metadata=None, subdomain_data=None, subdomain_id=None):
        if name not in ("dx", "ds"):
            raise NotImplementedError("cell measure dx and exterior-facet measure ds")
        self.name, self.domain = name, domain
        self.subdomain_data, self.subdomain_id = subdomain_data, subdomain_id  # ds: facet tags + the tags integrated over
        self.degree = None if not metadata else metadata.get("quadrature_degree")

    def __rmul__(self, o):
        return Form([Integral(as_expr(o), self)])

    def __call__(self, domain=None, metadata=None):
        """ufl.dx(domain=mesh) (intersecting_constraints_dolfinx.py:32)"""
        return Measure(self.name, domain=domain, metadata=metadata)


dx = Measure("dx")


@dataclass
class Integral:
    integrand: Expr
    measure: Measure
    scale: float = 1.0


class Form:
    def __init__(self, integrals):
        self.integrals = list(integrals)

    def __add__(self, o):
        if not isinstance(o, Form):
            return NotImplemented
        return Form(self.integrals + o.integrals)

    def __neg__(self):
        return Form([Integral(i.integrand, i.measure, -i.scale) for i in self.integrals])

    def __sub__(self, o):
        if not isinstance(o, Form):
            return NotImplemented
        return self + (-o)

    def __rmul__(self, c):
        c = as_expr(c)
        return Form([Integral(Product(c, i.integrand), i.measure, i.scale) for i in self.integrals])


@dataclass
class Derivative:
    """ufl.derivative(F, sol) (obstacle_pg.py:125): the exact Jacobian of the residual form."""
    form: object
    u: fem.Function


def TestFunction(V):
    """ufl.TestFunction(Z) of a mixed space: use ufl.split(z_test) for its components (intersecting_constraints_dolfinx.py:21-22)."""
    return _ArgumentTuple(V)


class _ArgumentTuple:
    def __init__(self, V):
        self.V = V
        self.parts = TestFunctions(V)


_split_function = split


def split(obj):  # noqa: F811 - ufl.split of a Function or of a TestFunction
    """ufl.split(sol) (obstacle_pg.py:88-89) / ufl.split(TestFunction(Z))."""
    if isinstance(obj, _ArgumentTuple):
        return obj.parts
    return _split_function(obj)


def _gateaux(e, u, test):
    """d/d eps e(u + eps test)|_0 for the expression subset: linear in the components of `u` replaced by the test functions,
    product / inner rules, grad linear, exp' = exp, sqrt' = 1/(2 sqrt)."""
    if isinstance(e, Number):
        return None
    if isinstance(e, Terminal):
        return test[e.index] if (e.kind == "component" and e.obj is u) else None
    if isinstance(e, Sum):
        a, b = _gateaux(e.a, u, test), _gateaux(e.b, u, test)
        return a if b is None else b if a is None else Sum(a, b)
    if isinstance(e, Scaled):
        a = _gateaux(e.a, u, test)
        return None if a is None else Scaled(e.c, a)
    if isinstance(e, (Product, Inner)):
        mk = Product if isinstance(e, Product) else Inner
        da, db = _gateaux(e.a, u, test), _gateaux(e.b, u, test)
        parts = ([mk(da, e.b)] if da is not None else []) + ([mk(e.a, db)] if db is not None else [])
        return None if not parts else parts[0] if len(parts) == 1 else Sum(parts[0], parts[1])
    if isinstance(e, Division):
        if _gateaux(e.b, u, test) is not None:
            raise NotImplementedError("derivative of a quotient whose denominator depends on the unknown")
        a = _gateaux(e.a, u, test)
        return None if a is None else Division(a, e.b)
    if isinstance(e, Grad):
        a = _gateaux(e.a, u, test)
        return None if a is None else Grad(a)
    if isinstance(e, Func):
        if e.name == "exp":
            a = _gateaux(e.args[0], u, test)
            return None if a is None else Product(e, a)
        if all(_gateaux(a, u, test) is None for a in e.args):
            return None
        raise NotImplementedError(f"derivative of {e.name}(...) with respect to the unknown")
    raise TypeError(type(e).__name__)


def derivative(F, u, du=None):
    """ufl.derivative.  Of a residual form with respect to the unknown (obstacle_pg.py:125): a marker - the HIP families assemble
    their exact Jacobians themselves.  Of an ENERGY functional with a TestFunction direction (intersecting_constraints_dolfinx.py:48,
    `derivative(E, z, z_test)`): the Gateaux derivative, symbolically, as a Form."""
    if isinstance(du, _ArgumentTuple):
        out = []
        for it in F.integrals:
            d = _gateaux(it.integrand, u, du.parts)
            if d is not None:
                out.append(Integral(d, it.measure, it.scale))
        return Form(out)
    return Derivative(F, u)


# ---------------------------------------------------------------------------------------------------------------------
# canonical form: {(measure degree, scalar factors, vector pair) -> coefficient}
# ---------------------------------------------------------------------------------------------------------------------


def _fmt(c):
    return repr(float(c))


class _Canon:
    def __init__(self, name_of):
        self.name_of = name_of  # Terminal -> (name, is_spatially_constant)

    # a monomial is (coef, tuple(sorted scalar factor names), vector factor name | None)
    def expand(self, e):
        if isinstance(e, Number):
            return [(e.v, (), None)]
        if isinstance(e, Terminal):
            name, _ = self.name_of(e)
            return [(1.0, (), name)] if e.rank else [(1.0, (name,), None)]
        if isinstance(e, Sum):
            return self.expand(e.a) + self.expand(e.b)
        if isinstance(e, Scaled):
            return [(e.c * c, s, v) for c, s, v in self.expand(e.a)]
        if isinstance(e, Product):
            out = []
            for ca, sa, va in self.expand(e.a):
                for cb, sb, vb in self.expand(e.b):
                    out.append((ca * cb, tuple(sorted(sa + sb)), va if va is not None else vb))
            return out
        if isinstance(e, Division):
            den = self.combine(self.expand(e.b))
            if len(den) == 1 and den[0][1] == () and den[0][2] is None:
                return [(c / den[0][0], s, v) for c, s, v in self.expand(e.a)]
            d = "1/(" + self.text(den) + ")"
            return [(c, tuple(sorted(s + (d,))), v) for c, s, v in self.expand(e.a)]
        if isinstance(e, Inner):
            out = []
            for ca, sa, va in self.expand(e.a):
                for cb, sb, vb in self.expand(e.b):
                    s = sa + sb
                    if va is not None:
                        if "I" in (va, vb):  # inner(I, X) = tr(X)
                            other = vb if va == "I" else va
                            core = other[4:-1] if other.startswith("sym(") else other
                            if not core.startswith("grad("):
                                raise NotImplementedError(f"inner(Identity, {other})")
                            s = s + ("div(" + core[5:-1] + ")",)
                        else:
                            s = s + ("inner(" + ",".join(sorted((va, vb))) + ")",)
                    out.append((ca * cb, tuple(sorted(s)), None))
            return out
        if isinstance(e, Grad):
            out = []
            for c, s, vslot in self.expand(e.a):
                if vslot is not None:  # gradient of a vector field: a tensor slot
                    if any(not self._const(f) for f in s):
                        raise NotImplementedError("grad of a product of fields")
                    out.append((c, s, "grad(" + vslot + ")"))
                    continue
                varying = [f for f in s if not self._const(f)]
                if len(varying) != 1:
                    raise NotImplementedError("grad of a product of fields or of a constant")
                rest = tuple(f for f in s if self._const(f))
                out.append((c, rest, "grad(" + varying[0] + ")"))
            return out
        if isinstance(e, Sym):
            return [(c, s, "sym(" + v + ")") for c, s, v in self.expand(e.a)]
        if isinstance(e, Tr):  # tr(grad(u)) = div(u); tr(sym(grad(u))) = div(u); tr(I) = dim
            out = []
            for c, s, v in self.expand(e.a):
                if v == "I":
                    raise NotImplementedError("tr(Identity)")
                core = v[4:-1] if v.startswith("sym(") else v
                if not core.startswith("grad("):
                    raise NotImplementedError(f"tr({v})")
                out.append((c, tuple(sorted(s + ("div(" + core[5:-1] + ")",))), None))
            return out
        if isinstance(e, IdentityTensor):
            return [(1.0, (), "I")]
        if isinstance(e, Func):
            return [(1.0, (e.name + "(" + ",".join(self.text(self.combine(self.expand(a))) for a in e.args) + ")",), None)]
        raise TypeError(f"unsupported expression node {type(e).__name__}")

    def _const(self, factor_name):
        return factor_name in self._const_names

    def combine(self, monos):
        acc = {}
        for c, s, v in monos:
            acc[(s, v)] = acc.get((s, v), 0.0) + c
        return sorted(((c, s, v) for (s, v), c in acc.items() if c != 0.0), key=lambda m: (m[1], m[2] or ""))

    def text(self, monos):
        return "+".join(_fmt(c) + "*" + "*".join(s + ((v,) if v else ())) for c, s, v in monos) or "0"

    def form(self, F: Form):
        acc = {}
        for it in F.integrals:
            if it.integrand.rank:
                raise ValueError("the integrand of a form must be scalar")
            for c, s, v in self.expand(it.integrand):
                k = (it.measure.degree, s) if it.measure.name == "dx" else ((it.measure.name, it.measure.degree), s)
                acc[k] = acc.get(k, 0.0) + it.scale * c
        return {k: c for k, c in acc.items() if c != 0.0}


def canonical(F: Form, names: dict):
    """names: Terminal.key() -> (name, spatially constant?).  Returns {(degree, factors): coefficient}."""
    cn = _Canon(lambda t: names[t.key()])
    cn._const_names = {n for n, const in names.values() if const}
    return cn.form(F)


def terminals(F: Form):
    seen, out = set(), []

    def walk(e):
        if isinstance(e, Terminal):
            if e.key() not in seen:
                seen.add(e.key())
                out.append(e)
        for a in ("a", "b"):
            if hasattr(e, a) and isinstance(getattr(e, a), Expr):
                walk(getattr(e, a))
        for a in getattr(e, "args", ()):
            walk(a)

    for it in F.integrals:
        walk(it.integrand)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# family templates and matching
# ---------------------------------------------------------------------------------------------------------------------


def _role(name, rank=0):
    return Terminal("role", name, 0, rank)


def _template_names(F, const_roles):
    return {t.key(): (t.obj, t.obj in const_roles) for t in terminals(F)}


def _obstacle_template(degree, roles_present):
    """The residual of obstacle_pg.py:116-124 in terms of roles (without the role f: the source term left out)."""
    u, psi, psi_k, v, w = (_role(n) for n in ("u", "psi", "psi_k", "v", "w"))
    alpha, f, phi = _role("alpha"), _role("f"), _role("phi")
    dx = Measure("dx", metadata={"quadrature_degree": degree})
    F = alpha * inner(grad(u), grad(v)) * dx + psi * v * dx + u * w * dx - exp(psi) * w * dx - phi * w * dx - psi_k * v * dx
    if "f" in roles_present:
        F = F - alpha * f * v * dx
    return F


def _thermoforming_template(degree, roles_present):
    """The residual of thermoforming_dolfinx.py:36-67 in terms of roles; with the role eps the form the reference
    differentiates for its modified Jacobian (:69-71)."""
    u, T, psi, psi_prev, v, q, w = (_role(n) for n in ("u", "T", "psi", "psi_prev", "v", "q", "w"))
    alpha, beta, f, bound0, bound1, eps = (_role(n) for n in ("alpha", "beta", "f", "bound0", "bound1", "eps"))
    x, y = _role("x"), _role("y")
    dx = Measure("dx", metadata={"quadrature_degree": degree} if degree is not None else None)

    def g(s):
        return conditional(lt(s, bound0), 1, conditional(lt(s, bound1), 1 - s / bound1, 0))

    Phi0 = 1 - 2 * max_value(abs(x - 0.5), abs(y - 0.5))
    xi = sin(pi * x) * sin(pi * y)
    F = alpha * inner(grad(u), grad(v)) * dx + inner(psi, v) * dx - alpha * inner(f, v) * dx - inner(psi_prev, v) * dx
    F += inner(grad(T), grad(q)) * dx + beta * inner(T, q) * dx - inner(g(exp(-psi)), q) * dx
    F += inner(u, w) * dx + inner(exp(-psi), w) * dx - inner(Phi0 + xi * T, w) * dx
    if "eps" in roles_present:
        F = F - eps / alpha * inner(grad(psi), grad(w)) * dx
    return F


def _gradient_constraint_template(degree, roles_present):
    """The residual of gradient_constraint_dolfinx.py:100-107 in terms of roles: u scalar, psi / psi0 / w vector-valued."""
    u, v = _role("u"), _role("v")
    psi, psi0, w = _role("psi", 1), _role("psi0", 1), _role("w", 1)
    alpha, f, phi = _role("alpha"), _role("f"), _role("phi")
    dx = Measure("dx", metadata={"quadrature_degree": degree})
    F = alpha * inner(grad(u), grad(v)) * dx + inner(psi, grad(v)) * dx - alpha * inner(f, v) * dx - inner(psi0, grad(v)) * dx
    F += inner(grad(u), w) * dx - phi * (1 / sqrt(1 + dot(psi, psi))) * dot(psi, w) * dx
    return F


def _intersecting_template(degree, roles_present):
    """The residual of intersecting_constraints_dolfinx.py:47-58 in terms of roles: alpha dE/dz + example 01's latent rows for psi0
    + example 06's for psi (compose(), below); phi0 and phi are DATA - expressions of the coordinates, lifted by lift_data()."""
    u, psi0, v, w0, psi0_iter = (_role(n) for n in ("u", "psi0", "v", "w0", "psi0_iter"))
    psi, w, psi_iter = _role("psi", 1), _role("w", 1), _role("psi_iter", 1)
    alpha, c, phi0, phi = (_role(n) for n in ("alpha", "c", "phi0", "phi"))
    dx = Measure("dx", metadata={"quadrature_degree": degree} if degree is not None else None)
    primal = alpha * inner(grad(u), grad(v)) * dx
    if "c" in roles_present:
        primal = primal + alpha * c * v * dx
    return compose(primal, [("exp", u, psi0, psi0_iter, v, w0, phi0, dx), ("hellinger", u, psi, psi_iter, v, w, phi, dx)])


@dataclass
class GradientConstraintSpec:
    """Example 06 (gradient_constraint_dolfinx.py:38-107): sol = (u, psi) in [P_k, (P_{k-1})^2], previous iterate w0, bound phi and
    source f as Functions of the collapsed primal space."""
    sol: fem.Function
    w0: fem.Function
    alpha: fem.Constant
    phi: fem.Function
    f: fem.Function
    quadrature_degree: int


@dataclass
class ThermoformingSpec:
    """Example 05 (thermoforming_dolfinx.py:28-71): s = (u, T, psi) in [P1]^3, g with knees bound0 < bound1, eps of the
    modified Jacobian (None: exact derivative)."""
    s: fem.Function
    s_prev: fem.Function
    alpha: fem.Constant
    beta: fem.Constant
    f: fem.Constant
    bound0: fem.Constant
    bound1: fem.Constant
    eps: fem.Constant | None
    quadrature_degree: int | None


@dataclass
class IntersectingSpec:
    """Example 08 (intersecting_constraints_dolfinx.py:13-58): z = (u, psi0, psi) in [P1, P1, (P1)^gdim], previous proximal iterate
    z_iter, the Constant c of the energy (None: absent), obstacle phi0 and gradient bound phi as DataExpressions (sampled at the
    quadrature points at every solve: their Constants - `phic` - stay live)."""
    z: fem.Function
    z_iter: fem.Function
    alpha: fem.Constant
    c: fem.Constant | None
    phi0: "DataExpression"
    phi: "DataExpression"
    quadrature_degree: int | None


# ---------------------------------------------------------------------------------------------------------------------
# data expressions: sub-expressions of the coordinates and Constants only (phi0, phi of example 08)
# ---------------------------------------------------------------------------------------------------------------------
class DataExpression:
    """A maximal sub-expression of a form that depends on the spatial coordinates (and Constants) only.  The families treat it as a
    coefficient; the host evaluates it where the kernels need it (FFCx would inline it into the generated element kernel)."""

    def __init__(self, expr: Expr, mesh):
        self.expr, self.mesh = expr, mesh

    def __call__(self, x):
        """x: (gdim, npts) like a dolfinx interpolation callable -> (npts,)"""
        x = np.asarray(x, dtype=np.float64)
        with np.errstate(all="ignore"):  # both branches of a conditional are evaluated
            v = _evaluate(self.expr, x)
        return np.broadcast_to(np.asarray(v, dtype=np.float64), x.shape[1:]).copy()


def _evaluate(e, x):
    if isinstance(e, Number):
        return e.v
    if isinstance(e, Terminal):
        if e.kind == "coordinate":
            return x[e.index]
        if e.kind == "constant" and e.rank == 0:
            return float(e.obj.value)
        raise NotImplementedError(f"a {e.kind} terminal inside a data expression")
    if isinstance(e, Sum):
        return _evaluate(e.a, x) + _evaluate(e.b, x)
    if isinstance(e, Scaled):
        return e.c * _evaluate(e.a, x)
    if isinstance(e, Product):
        return _evaluate(e.a, x) * _evaluate(e.b, x)
    if isinstance(e, Division):
        return _evaluate(e.a, x) / _evaluate(e.b, x)
    if isinstance(e, Func):
        a = [_evaluate(q, x) for q in e.args]
        if e.name == "conditional":
            return np.where(a[0], a[1], a[2])
        fn = {"exp": np.exp, "sqrt": np.sqrt, "sin": np.sin, "abs": np.abs, "max_value": np.maximum, "lt": np.less,
              "le": np.less_equal, "gt": np.greater, "ge": np.greater_equal}.get(e.name)
        if fn is None:
            raise NotImplementedError(f"{e.name} in a data expression")
        return fn(*a)
    raise NotImplementedError(f"{type(e).__name__} in a data expression")


def lift_data(F: Form):
    """-> (form with every maximal coordinate-dependent data sub-expression replaced by a 'coefficient' Terminal whose object is a
    DataExpression, list of those DataExpressions).  Identical sub-expression OBJECTS share one placeholder."""
    kinds = {}

    def data_kind(e):  # 0: not data; 1: data without a coordinate (numbers, Constants); 2: data that varies in space
        k = kinds.get(id(e))
        if k is not None:
            return k
        if isinstance(e, Number):
            k = 1
        elif isinstance(e, Terminal):
            k = 2 if e.kind == "coordinate" else 1 if (e.kind == "constant" and e.rank == 0) else 0
        elif isinstance(e, (Sum, Product, Division)):
            ka, kb = data_kind(e.a), data_kind(e.b)
            k = 0 if 0 in (ka, kb) else max(ka, kb)
        elif isinstance(e, Scaled):
            k = data_kind(e.a)
        elif isinstance(e, Func):
            ks = [data_kind(a) for a in e.args]
            k = 0 if 0 in ks else max(ks)
        else:
            k = 0
        kinds[id(e)] = k
        return k

    lifted, by_id = [], {}

    def walk(e):
        if data_kind(e) == 2:
            t = by_id.get(id(e))
            if t is None:
                mesh = next(q.obj for q in _walk_terminals(e) if q.kind == "coordinate")
                d = DataExpression(e, mesh)
                lifted.append(d)
                t = by_id[id(e)] = Terminal("coefficient", d)
            return t
        if isinstance(e, (Sum, Product, Division, Inner)):
            return type(e)(walk(e.a), walk(e.b))
        if isinstance(e, Scaled):
            return Scaled(e.c, walk(e.a))
        if isinstance(e, (Grad, Sym, Tr)):
            return type(e)(walk(e.a))
        if isinstance(e, Func):
            return Func(e.name, *[walk(a) for a in e.args])
        return e

    return Form([Integral(walk(it.integrand), it.measure, it.scale) for it in F.integrals]), lifted


def _walk_terminals(e):
    if isinstance(e, Terminal):
        yield e
    for a in ("a", "b"):
        if isinstance(getattr(e, a, None), Expr):
            yield from _walk_terminals(getattr(e, a))
    for a in getattr(e, "args", ()):
        yield from _walk_terminals(a)


# ---------------------------------------------------------------------------------------------------------------------
# latent-variable rows: the building blocks the family templates share, and their COMPOSITION
# ---------------------------------------------------------------------------------------------------------------------
def exp_latent_rows(u, psi, psi_prev, v, w, phi, dx):
    """Rows an obstacle-type constraint u >= phi adds (obstacle_pg.py:118-123): (psi - psi_prev, v) in the primal equation and
    the latent equation (u, w) - (exp(psi), w) - (phi, w)."""
    return inner(psi, v) * dx - inner(psi_prev, v) * dx + inner(u, w) * dx - inner(exp(psi), w) * dx - inner(phi, w) * dx


def hellinger_latent_rows(u, psi, psi_prev, v, w, phi, dx):
    """Rows a gradient bound |grad u| <= phi adds (gradient_constraint_dolfinx.py:101-107): (psi - psi_prev, grad v) in the primal
    equation and the latent equation (grad u, w) - (phi psi / sqrt(1 + psi.psi), w)."""
    return (inner(psi, grad(v)) * dx - inner(psi_prev, grad(v)) * dx + inner(grad(u), w) * dx
            - inner(phi * psi / sqrt(1 + dot(psi, psi)), w) * dx)


def compose(primal_rows: Form, constraints):
    """Residual of a problem with SEVERAL latent variables on one primal field - example 08's structure
    (intersecting_constraints_dolfinx.py:47-58: an obstacle AND a gradient bound): the primal rows plus, per constraint,
    ("exp" | "hellinger", u, psi, psi_prev, v, w, phi, dx).  The forms of examples 01 and 06 are compose() with one constraint."""
    F = primal_rows
    for kind, *args in constraints:
        F = F + {"exp": exp_latent_rows, "hellinger": hellinger_latent_rows}[kind](*args)
    return F


def canonical_by_position(F: Form):
    """Canonical monomials of a form with every terminal named by what it IS (component i of function k, test function i,
    constant k, ...) in order of first appearance of the underlying objects: two forms over the same objects compare equal iff
    they are equal as polynomials in their terminals."""
    objs, names = [], {}

    def oid(o):
        for k, q in enumerate(objs):
            if q is o:
                return k
        objs.append(o)
        return len(objs) - 1

    for t in terminals(F):
        tag = {"component": "fn", "argument": "test", "constant": "c", "quadrature": "q", "coefficient": "coef", "coordinate": "x",
               "role": "role", "field": "field"}[t.kind]
        names[t.key()] = (f"{tag}{oid(t.obj)}[{t.index}]", t.kind == "constant")
    return canonical(F, names), objs


def forms_equal(F: Form, G: Form, tol=1e-14):
    """Equality of two forms over the same Python objects (Functions, Constants, ...) as polynomials in their terminals."""
    both = Form(F.integrals + (-G).integrals)
    c, _ = canonical_by_position(both)
    return all(abs(v) <= tol for v in c.values())


# name, components of the unknown, of the previous iterate, test functions, roles of Constants (required, optional),
# number of quadrature-space coefficients (role phi), template
_FAMILIES = [
    dict(name="obstacle (example 01)", comps=("u", "psi"), prev=("u_k", "psi_k"), args=("v", "w"), required=("alpha",),
         optional=("f",), quads=1, needs_degree=True, template=_obstacle_template,
         text="alpha*inner(grad(u),grad(v)) + psi*v + u*w - exp(psi)*w - phi*w - alpha*f*v - psi_k*v"),
    dict(name="thermoforming QVI (example 05)", comps=("u", "T", "psi"), prev=("u_prev", "T_prev", "psi_prev"),
         args=("v", "q", "w"), required=("alpha", "beta", "f", "bound0", "bound1"), optional=("eps",), quads=0,
         needs_degree=False, template=_thermoforming_template, text="the residual of thermoforming_dolfinx.py:62-67"),
    # vector latent variable: component ranks (0, 1); phi and f are coefficient Functions of the collapsed primal space
    dict(name="gradient constraint (example 06)", comps=("u", "psi"), prev=("u0", "psi0"), args=("v", "w"), ranks=(0, 1),
         required=("alpha",), optional=(), coefs=("phi", "f"), quads=0, needs_degree=True, template=_gradient_constraint_template,
         text="alpha*inner(grad(u),grad(v)) + inner(psi,grad(v)) - alpha*f*v - inner(psi0,grad(v)) + inner(grad(u),w) "
              "- phi*dot(psi,w)/sqrt(1+dot(psi,psi))"),
    # two latent variables on one primal field; phi0 and phi are data expressions lifted to coefficients (compile_form)
    dict(name="intersecting constraints (example 08)", comps=("u", "psi0", "psi"), prev=("u_iter", "psi0_iter", "psi_iter"),
         args=("v", "w0", "w"), ranks=(0, 0, 1), required=("alpha",), optional=("c",), coefs=("phi0", "phi"), quads=0,
         needs_degree=False, template=_intersecting_template,
         text="alpha*(inner(grad(u),grad(v)) + c*v) + exp_latent_rows(u, psi0; phi0) + hellinger_latent_rows(u, psi; phi)"),
]


def _describe(diff):
    return "; ".join(f"{_fmt(c)} * {' * '.join(s)} [quadrature degree {d}]" for (d, s), c in sorted(diff.items(), key=str))


def _match(F: Form, u: fem.Function):
    """-> (family, {role: object}, degree) or raises NotImplementedError."""
    if not isinstance(F, Form):
        raise TypeError("F must be a Form")
    terms = terminals(F)
    V = u.function_space
    comps = [t for t in terms if t.kind == "component"]
    others = []
    for t in comps:
        if t.obj is not u and t.obj not in others:
            others.append(t.obj)
    args = [t for t in terms if t.kind == "argument"]
    if any(t.obj is not V for t in args):
        raise ValueError("test functions must come from the unknown's function space")
    consts = [t for t in terms if t.kind == "constant"]
    quads = [t for t in terms if t.kind == "quadrature"]
    coords = [t for t in terms if t.kind == "coordinate"]
    coefs = [t for t in terms if t.kind == "coefficient"]
    ranks = tuple(_comp_rank(V, i) for i in range(V.ncomp))
    degrees = {it.measure.degree for it in F.integrals}
    if len(degrees) != 1:
        raise NotImplementedError("all integrals must use one measure")
    degree = degrees.pop()
    problems = []
    for fam in _FAMILIES:
        nreq, nall = len(fam["required"]), len(fam["required"]) + len(fam["optional"])
        if (V.ncomp != len(fam["comps"]) or len(others) != 1 or len(quads) != fam["quads"] or not nreq <= len(consts) <= nall
                or (fam["needs_degree"] and degree is None) or ranks != fam.get("ranks", (0,) * V.ncomp)
                or len(coefs) != len(fam.get("coefs", ()))):
            continue
        fixed = {}
        for t in comps:
            fixed[t.key()] = ((fam["comps"] if t.obj is u else fam["prev"])[t.index], False)
        for t in args:
            fixed[t.key()] = (fam["args"][t.index], False)
        for t in quads:
            fixed[t.key()] = ("phi", False)
        for t in coords:
            fixed[t.key()] = ("xyz"[t.index], False)
        best = None
        templates = {}
        for cperm in itertools.permutations(fam.get("coefs", ())):  # which coefficient Function plays which role
          for perm in itertools.permutations(fam["required"] + fam["optional"], len(consts)):
            if not set(fam["required"]) <= set(perm):
                continue
            present = frozenset(perm)
            if present not in templates:
                T = fam["template"](degree, present)
                templates[present] = canonical(T, _template_names(T, set(fam["required"] + fam["optional"])))
            template = templates[present]
            names = dict(fixed)
            for t, r in zip(consts, perm):
                names[t.key()] = (r, True)
            for t, r in zip(coefs, cperm):
                names[t.key()] = (r, False)
            got = canonical(F, names)
            diff = {k: got.get(k, 0.0) - template.get(k, 0.0) for k in set(got) | set(template)}
            diff = {k: c for k, c in diff.items() if abs(c) > 1e-14}
            if not diff:
                roles = {r: t.obj for t, r in zip(consts, perm)}
                roles.update({r: t.obj for t, r in zip(coefs, cperm)})
                roles.update(unknown=u, previous=others[0], quads=[t.obj for t in quads])
                return fam, roles, degree
            if best is None or len(diff) < len(best):
                best = diff
        problems.append(f"{fam['name']}: terms that differ from {fam['text']}: " + _describe(best or {}))
    raise NotImplementedError("the form matches no problem family implemented in HIP" +
                              ("".join("\n  " + p for p in problems) if problems else
                               " (expected the mixed unknown and its previous iterate of example 01, 05 or 06 with their Constants)"))


def compile_form(F: Form, u: fem.Function, J=None):
    """Match a residual form (and, if given, the Jacobian J = derivative(G, u)) against the problem families the HIP path
    implements.  Returns the family's declarative description: problem.ObstacleResidual (example 01, J must be the exact
    derivative) or ThermoformingSpec (example 05, G may carry the -eps/alpha (grad psi, grad w) modification)."""
    from .problem import ObstacleResidual

    try:
        fam, r, degree = _match(F, u)
    except NotImplementedError as first:
        # expressions of the coordinates that no family spells out (example 08's phi0, phi): lift them to coefficients, match again
        Fl, lifted = lift_data(F) if isinstance(F, Form) else (F, [])
        if not lifted:
            raise
        try:
            fam, r, degree = _match(Fl, u)
        except NotImplementedError:
            raise first from None
    if fam["name"].startswith("intersecting"):
        if J is not None and getattr(J, "form", None) is not F:
            raise NotImplementedError("example 08 passes no Jacobian form: NonlinearProblem differentiates F "
                                      "(intersecting_constraints_dolfinx.py:75-77)")
        return IntersectingSpec(u, r["previous"], r["alpha"], r.get("c"), r["phi0"], r["phi"], degree)
    G = None
    if J is not None:
        if not (hasattr(J, "form") and hasattr(J, "u")) or J.u is not u:
            raise NotImplementedError("J must be derivative(G, u) of a form G with respect to the unknown")
        G = J.form
    V = u.function_space
    if fam["name"].startswith("gradient constraint"):
        if G is not None and G is not F:
            raise NotImplementedError("example 06 passes no Jacobian form: NonlinearProblem differentiates F (gradient_constraint_dolfinx.py:113)")
        return GradientConstraintSpec(u, r["previous"], r["alpha"], r["phi"], r["f"], degree)
    if fam["name"].startswith("obstacle"):
        if G is not None and G is not F:
            famG, rG, _ = _match(G, u)
            if famG is not fam or any(rG.get(k) is not r.get(k) for k in ("alpha", "f", "previous")) or rG["quads"] != r["quads"]:
                raise NotImplementedError("only J = derivative(F, u) (the exact Jacobian) is supported for the obstacle family")
        return ObstacleResidual(u, r["previous"], r["alpha"], r.get("f") or fem.Constant(V.mesh, 0.0), r["quads"][0], degree)
    eps = None
    if "eps" in r:
        raise NotImplementedError("the eps modification belongs in the Jacobian form, not in the residual")
    if G is not None and G is not F:
        famG, rG, _ = _match(G, u)
        same = all(rG.get(k) is r.get(k) for k in ("alpha", "beta", "f", "bound0", "bound1", "previous"))
        if famG is not fam or not same:
            raise NotImplementedError("J must be the derivative of F or of F - eps/alpha*inner(grad(psi),grad(w))*dx")
        eps = rG.get("eps")
    return ThermoformingSpec(u, r["previous"], r["alpha"], r["beta"], r["f"], r["bound0"], r["bound1"], eps, degree)


# ---------------------------------------------------------------------------------------------------------------------
# example 02: blocked residual with a facet latent variable (signorini_dolfinx.py:211-252)
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class SignoriniSpec:
    u: fem.Function
    psi: fem.Function
    psi_k: fem.Function
    alpha: fem.Constant
    mu: float
    lmbda: float
    gap: float
    contact_facets: object  # (nf, 3) vertex triples the ds measure integrates over
    quadrature_degree: int


def compile_signorini(F: Form, unknowns):
    """Match the blocked residual of signorini_dolfinx.py:244-249 - alpha (sigma(u), eps(v)) dx - alpha (f, v) dx
    - (psi - psi_k, v.n_g) ds + (u.n_g, w) ds + (exp(psi), w) ds - (g, w) ds with sigma = 2 mu eps + lambda tr(grad u) I,
    g = x_last + Constant(-gap) - against the family's template.  mu and lambda are plain numbers in the reference: they are READ
    OFF the coefficients of the two elasticity monomials; everything else must agree term by term."""
    u, psi = unknowns
    terms = terminals(F)
    fields = [t for t in terms if t.kind == "field"]
    args = [t for t in terms if t.kind == "argument"]
    consts = [t for t in terms if t.kind == "constant"]
    coords = [t for t in terms if t.kind == "coordinate"]
    Q = args[0].obj if args else None
    if not isinstance(Q, MixedFunctionSpace) or any(t.obj is not Q for t in args) or Q.ncomp != 2:
        raise NotImplementedError("test functions must come from MixedFunctionSpace(V, W)")
    others = [t.obj for t in fields if t.obj is not u and t.obj is not psi]
    if len(others) != 1 or others[0].function_space is not psi.function_space:
        raise NotImplementedError("expected exactly one more Function of the latent space (the previous iterate psi_k)")
    psi_k = others[0]
    dim = u.function_space.dim
    vec = [t for t in consts if t.rank == 1]
    sca = [t for t in consts if t.rank == 0]
    if len(vec) != 2 or len(sca) != 2 or len(coords) != 1 or coords[0].index != dim - 1:
        raise NotImplementedError("expected the vector Constants n_g and f, the scalar Constants alpha and -gap and the last coordinate")
    meas = {(it.measure.name, it.measure.degree) for it in F.integrals}
    dsm = [it.measure for it in F.integrals if it.measure.name == "ds"]
    if not dsm or any(m.subdomain_data is not dsm[0].subdomain_data or m.subdomain_id != dsm[0].subdomain_id for m in dsm):
        raise NotImplementedError("one ds measure over the contact tags")
    ds_deg = dsm[0].degree
    if meas != {("dx", None), ("ds", ds_deg)} or ds_deg is None:
        raise NotImplementedError("integrals over ufl.dx(domain=mesh) and one ds measure with a quadrature degree")
    base = {}
    for t in fields:
        base[t.key()] = ({id(u): "u", id(psi): "psi", id(psi_k): "psi_k"}[id(t.obj)], False)
    for t in args:
        base[t.key()] = (("v", "w")[t.index], False)
    base[coords[0].key()] = ("xg", False)
    # template in roles, with unit elasticity coefficients (their values are read off below)
    ru, rv, rn, rf = _role("u", 1), _role("v", 1), _role("n_g", 1), _role("f", 1)
    rpsi, rpsik, rw, ralpha, rgap, rx = (_role(n) for n in ("psi", "psi_k", "w", "alpha", "mgap", "xg"))
    tdx, tds = Measure("dx"), Measure("ds", metadata={"quadrature_degree": ds_deg})
    eps = lambda a: sym(grad(a))  # noqa: E731
    T = (ralpha * inner(eps(ru), eps(rv)) * tdx + ralpha * tr(grad(ru)) * tr(grad(rv)) * tdx - ralpha * inner(rf, rv) * tdx
         - inner(rpsi - rpsik, dot(rv, rn)) * tds + inner(dot(ru, rn), rw) * tds + inner(exp(rpsi), rw) * tds - inner(rx + rgap, rw) * tds)
    tmpl = canonical(T, _template_names(T, {"alpha", "mgap", "n_g", "f"}))
    k_eps = next(k for k in tmpl if any(f.startswith("inner(sym(") for f in k[1]))
    k_div = next(k for k in tmpl if sum(f.startswith("div(") for f in k[1]) == 2)
    best = None
    for vperm in itertools.permutations(("n_g", "f")):
        for sperm in itertools.permutations(("alpha", "mgap")):
            names = dict(base)
            for t, r in zip(vec, vperm):
                names[t.key()] = (r, True)
            for t, r in zip(sca, sperm):
                names[t.key()] = (r, True)
            got = canonical(F, names)
            two_mu, lam = got.get(k_eps, 0.0), got.get(k_div, 0.0)
            diff = {k: got.get(k, 0.0) - tmpl.get(k, 0.0) for k in (set(got) | set(tmpl)) - {k_eps, k_div}}
            diff = {k: c for k, c in diff.items() if abs(c) > 1e-14}
            if not diff and two_mu > 0.0 and lam >= 0.0:
                roles = {r: t.obj for t, r in zip(vec, vperm)}
                roles.update({r: t.obj for t, r in zip(sca, sperm)})
                n_g, f = roles["n_g"].value, roles["f"].value
                e_last = np.zeros(dim)
                e_last[-1] = -1.0
                if not np.array_equal(np.asarray(n_g), e_last) or np.any(np.asarray(f) != 0.0):
                    raise NotImplementedError("HIP backend: n_g = -e_last and f = 0 (the reference's values, signorini_dolfinx.py:236-240)")
                m = dsm[0]
                ids = m.subdomain_id if isinstance(m.subdomain_id, (tuple, list)) else (m.subdomain_id,)
                facets = np.concatenate([m.subdomain_data.find(i) for i in ids])
                return SignoriniSpec(u, psi, psi_k, roles["alpha"], 0.5 * two_mu, lam, -float(roles["mgap"].value), facets, ds_deg)
            if best is None or len(diff) < len(best):
                best = diff
    raise NotImplementedError("the form is not the Signorini residual of signorini_dolfinx.py:244-249; terms that differ: " + _describe(best or {}))
