#!/usr/bin/env python3
"""Generate proximalgalerkin_amd/tables/quadrature.json — the ONE place quadrature tables live.

Both the CPU oracle (oracle/) and the HIP path (proximalgalerkin_amd/) read the JSON this script
writes, so their tables are bit-identical.

Tables
------
tri_deg6_12 : fully symmetric 12-point, degree-6 rule on the reference triangle
              {(x,y): x,y>=0, x+y<=1} (Dunavant 1985, Int. J. Numer. Meth. Engng 21:1129-1148,
              rule p=6).  The 15-digit published values are used only as a starting guess: the
              seven free parameters (w1,a1,w2,a2,w3,b3,c3) are re-solved here with mpmath from the
              seven S3-invariant moment equations up to degree 6, to 40 digits, then rounded to
              double.  So the table is *derivable* from this script alone.
              Weights sum to 1/2 (reference-triangle area), matching the convention of
              SURVEY.md App. A.2.

tri_deg10_gj36 : collapsed (Duffy) Gauss-Jacobi rule, 6 x 6 = 36 points, exact to degree 11 >= 10: Gauss-Jacobi(1,0) in
              the collapsing direction times Gauss-Legendre across - Basix's "gauss_jacobi" scheme with
              m = (degree+2)//2 points per direction.  Example 06 fixes quadrature_degree=10
              (/root/reference/examples/06_gradient_constraints/gradient_constraint_dolfinx.py:53); Basix's DEFAULT for
              that degree is a 25-point Xiao-Gimbutas table that cannot be derived offline, hence this derivable rule
              (parity with FEniCSx unpinned, oracle <-> HIP exact by construction).  Nodes are roots of the Jacobi /
              Legendre polynomials refined with mpmath, weights from the moment equations, 50 digits.

The reference fixes quadrature_degree=6 for every integral of example 01
(/root/reference/examples/01_obstacle_problem/obstacle_pg.py:106,115).  Basix's default table for
that degree is not available offline => parity with a real FEniCSx run is
"unpinned" until the table is confirmed; parity between oracle and HIP is exact by construction.
"""
import itertools
import json
import pathlib

import mpmath as mp

mp.mp.dps = 50


def moments_exact(p, q):
    """int_T x^p y^q dx dy on the reference triangle = p! q! / (p+q+2)!"""
    return mp.factorial(p) * mp.factorial(q) / mp.factorial(p + q + 2)


def orbit3(a):
    """S3 orbit of barycentric point (a,a,1-2a) -> 3 points (x,y)."""
    b = 1 - 2 * a
    return [(a, a), (a, b), (b, a)]


def orbit6(b, c):
    a = 1 - b - c
    pts = set(itertools.permutations((a, b, c)))
    return [(p[0], p[1]) for p in sorted(pts)]


def rule(params):
    w1, a1, w2, a2, w3, b3, c3 = params
    pts, wts = [], []
    for w, a in ((w1, a1), (w2, a2)):
        for p in orbit3(a):
            pts.append(p)
            wts.append(w)
    for p in orbit6(b3, c3):
        pts.append(p)
        wts.append(w3)
    return pts, wts


# S3-invariant test polynomials spanning invariants up to degree 6 (7 of them):
# use monomials x^p y^q with (p,q) chosen so the 7x7 system is non-singular.
MONOS = [(0, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (3, 3)]


def residual(*params):
    pts, wts = rule(params)
    out = []
    for p, q in MONOS:
        s = mp.mpf(0)
        for (x, y), w in zip(pts, wts):
            s += w * x**p * y**q
        out.append(s - moments_exact(p, q))
    return out


def gauss_jacobi_01(m, a):
    """m-point Gauss rule on [0,1] for the weight (1-t)^a, a in {0,1}: (nodes, weights) as mpf lists."""
    import numpy as np
    from scipy.special import roots_jacobi

    def jac(t):  # P_m^{(a,0)}(t) by its finite sum (mpmath's hypergeometric form fails to converge at t = 0)
        return sum(mp.binomial(m + a, m - s) * mp.binomial(m, s) * ((t - 1) / 2) ** s * ((t + 1) / 2) ** (m - s)
                   for s in range(m + 1))

    guess, _ = roots_jacobi(m, a, 0)
    nodes = [mp.findroot(jac, mp.mpf(float(g)) + mp.mpf("1e-20"), tol=1e-45) for g in guess]
    nodes = [(1 + t) / 2 for t in nodes]  # [-1,1] -> [0,1]; weight (1-t)^a keeps its form up to a constant
    A = mp.matrix(m, m)
    b = mp.matrix(m, 1)
    for k in range(m):
        for i, t in enumerate(nodes):
            A[k, i] = t**k
        b[k] = mp.quad(lambda t: (1 - t) ** a * t**k, [0, 1])
    w = mp.lu_solve(A, b)
    return nodes, [w[i] for i in range(m)]


def collapsed_rule(m):
    """x = xi, y = eta (1 - xi): int_T f = int_0^1 int_0^1 f (1-xi) d eta d xi."""
    xi, wxi = gauss_jacobi_01(m, 1)
    eta, weta = gauss_jacobi_01(m, 0)
    pts, wts = [], []
    for a, wa in zip(xi, wxi):
        for b, wb in zip(eta, weta):
            pts.append((a, b * (1 - a)))
            wts.append(wa * wb)
    return pts, wts


def main():
    # Dunavant p=6 published values (weights there sum to 1; halve for area 1/2)
    guess = [
        mp.mpf("0.116786275726379") / 2,
        mp.mpf("0.249286745170910"),
        mp.mpf("0.050844906370207") / 2,
        mp.mpf("0.063089014491502"),
        mp.mpf("0.082851075618374") / 2,
        mp.mpf("0.310352451033784"),
        mp.mpf("0.636502499121399"),
    ]
    sol = mp.findroot(residual, guess, tol=1e-40, maxsteps=50)
    params = [sol[i] for i in range(7)]
    pts, wts = rule(params)
    # verify every monomial up to degree 6
    worst = mp.mpf(0)
    for p in range(7):
        for q in range(7 - p):
            s = sum(w * x**p * y**q for (x, y), w in zip(pts, wts))
            worst = max(worst, abs(s - moments_exact(p, q)))
    assert worst < mp.mpf(10) ** (-35), worst
    for g, s in zip(guess, params):
        assert abs(g - s) < 1e-13, (g, s)  # we converged to Dunavant's rule, not another root

    table = {
        "tri_deg6_12": {
            "cell": "triangle",
            "degree": 6,
            "source": "Dunavant (1985) p=6, 12 points; re-solved to 40 digits by tools/make_quadrature_tables.py",
            "points": [[float(x), float(y)] for (x, y) in pts],
            "weights": [float(w) for w in wts],
        }
    }
    # The OTHER fully symmetric 12-point degree-6 rule with orbit structure [3, 3, 6], interior points and positive weights: the
    # global search of tools/quadrature_uniqueness.py (30 000 starts) finds exactly two admissible roots of the moment system,
    # Dunavant's and this one.  Selected with scheme="tri_deg6_12_b" (fem.QuadratureFunction, setup_problem); never the default.
    guess_b = [mp.mpf("0.08566656207648825"), mp.mpf("0.21942998254978308"), mp.mpf("0.04036554479651741"),
               mp.mpf("0.4801379641122133"), mp.mpf("0.02031727989683051"), mp.mpf("0.14161901592396486"),
               mp.mpf("0.8390092597147903")]
    sol_b = mp.findroot(residual, guess_b, tol=1e-40, maxsteps=50)
    params_b = [sol_b[i] for i in range(7)]
    pts_b, wts_b = rule(params_b)
    worst_b = mp.mpf(0)
    for p in range(7):
        for q in range(7 - p):
            s = sum(w * x**p * y**q for (x, y), w in zip(pts_b, wts_b))
            worst_b = max(worst_b, abs(s - moments_exact(p, q)))
    assert worst_b < mp.mpf(10) ** (-35), worst_b
    for g, sv in zip(guess_b, params_b):
        assert abs(g - sv) < 1e-10, (g, sv)
    assert min(wts_b) > 0 and all(x > 0 and y > 0 and x + y < 1 for x, y in pts_b)
    pts10, wts10 = collapsed_rule(6)
    worst10 = mp.mpf(0)
    for p in range(12):
        for q in range(12 - p):
            s = sum(w * x**p * y**q for (x, y), w in zip(pts10, wts10))
            worst10 = max(worst10, abs(s - moments_exact(p, q)))
    assert worst10 < mp.mpf(10) ** (-35), worst10
    table["tri_deg10_gj36"] = {
        "cell": "triangle",
        "degree": 10,
        "source": "collapsed Gauss-Jacobi(1,0) x Gauss-Legendre, 6 x 6 points (exact to degree 11); tools/make_quadrature_tables.py",
        "points": [[float(x), float(y)] for (x, y) in pts10],
        "weights": [float(w) for w in wts10],
    }
    # facet rule of example 02 (quadrature_degree 4 on the contact triangles, signorini_dolfinx.py:67-69,211-218): same
    # derivable family, m = (4+2)//2 = 3 points per direction, exact to degree 5
    pts4, wts4 = collapsed_rule(3)
    worst4 = mp.mpf(0)
    for p in range(6):
        for q in range(6 - p):
            s = sum(w * x**p * y**q for (x, y), w in zip(pts4, wts4))
            worst4 = max(worst4, abs(s - moments_exact(p, q)))
    assert worst4 < mp.mpf(10) ** (-35), worst4
    table["tri_deg4_gj9"] = {
        "cell": "triangle",
        "degree": 4,
        "source": "collapsed Gauss-Jacobi(1,0) x Gauss-Legendre, 3 x 3 points (exact to degree 5); tools/make_quadrature_tables.py",
        "points": [[float(x), float(y)] for (x, y) in pts4],
        "weights": [float(w) for w in wts4],
    }
    # vertex (trapezoidal) rule, exact to degree 1: with it the P1 system on a uniform right-triangle grid IS the 5-point
    # finite-difference LVPP system of obstacle_finite_difference.jl (lumped mass, lumped D) - oracle/fd_oracle.py
    table["tri_vertex_3"] = {
        "cell": "triangle",
        "degree": 1,
        "source": "vertex rule: the three vertices, weights 1/6 (exact to degree 1; mass lumping); tools/make_quadrature_tables.py",
        "points": [[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]],
        "weights": [1.0 / 6.0, 1.0 / 6.0, 1.0 / 6.0],
    }
    table["tri_deg6_12_b"] = {
        "cell": "triangle",
        "degree": 6,
        "scheme_only": True,  # never picked by (cell, degree): by name only
        "source": "the second admissible root of the [3,3,6] degree-6 moment system (tools/quadrature_uniqueness.py), solved to "
                  "40 digits by tools/make_quadrature_tables.py",
        "points": [[float(x), float(y)] for (x, y) in pts_b],
        "weights": [float(w) for w in wts_b],
    }
    out = pathlib.Path(__file__).resolve().parents[1] / "proximalgalerkin_amd" / "tables" / "quadrature.json"
    out.write_text(json.dumps(table, indent=1) + "\n")
    print("wrote", out, "max moment error", mp.nstr(worst, 3))


if __name__ == "__main__":
    main()
