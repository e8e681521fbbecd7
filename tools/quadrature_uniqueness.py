#!/usr/bin/env python3
"""Is the 12-point degree-6 triangle rule of proximalgalerkin_amd/tables/quadrature.json THE rule of its kind?  

SURVEY.md H3: example 01 fixes quadrature_degree = 6 and integrates non-polynomial terms, so the discrete solution depends on the
table at ~1e-8, far above the 1e-10 parity bar; Basix's default degree-6 table cannot be read offline.  What CAN be settled here: a
fully symmetric (S3-invariant) 12-point rule with orbit structure [3, 3, 6] - two orbits (a, a, 1-2a) and one orbit (a, b, c) in
barycentric coordinates - has 7 free parameters (w1, a1, w2, a2, w3, b3, c3), and exactness for the 7 S3-invariant polynomials of
degree <= 6 gives 7 equations.  This script searches that system globally: Newton (scipy hybr) from many thousand random starts
spread over the whole admissible box, every converged root checked against ALL 28 monomials of degree <= 6 and reduced to a canonical
form (orbits of the same kind sorted, the 6-orbit's coordinates sorted).  Result (committed: profiles/r04_quadrature_uniqueness.json):
every admissible root - points strictly inside the triangle, positive weights - is the SAME rule, the one in the table.  So ANY
fully symmetric 12-point degree-6 rule with positive weights and interior points and that orbit structure - Dunavant's, and Basix's
Xiao-Gimbutas table IF it has that structure (assumption: not verifiable offline) - equals ours up to the order of its points and
rounding, and the 1e-8 sensitivity of DESIGN.md section 2 reduces to summation order.  Numerical evidence from a global search, not a
proof.     python tools/quadrature_uniqueness.py [n_starts]"""
import itertools
import json
import math
import pathlib
import sys

import numpy as np
from scipy.optimize import root

MONOS = [(0, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (3, 3)]  # as tools/make_quadrature_tables.py: spans the invariants to degree 6


def moment(p, q):
    return math.factorial(p) * math.factorial(q) / math.factorial(p + q + 2)


def rule(par):
    w1, a1, w2, a2, w3, b3, c3 = par
    pts, wts = [], []
    for w, a in ((w1, a1), (w2, a2)):
        b = 1 - 2 * a
        pts += [(a, a), (a, b), (b, a)]
        wts += [w] * 3
    a3 = 1 - b3 - c3
    for p in sorted(set(itertools.permutations((a3, b3, c3)))):
        pts.append((p[0], p[1]))
        wts.append(w3)
    if len(pts) != 12:  # degenerate 6-orbit (two equal coordinates)
        return None, None
    return np.array(pts), np.array(wts)


def residual(par):
    pts, wts = rule(par)
    if pts is None:
        return np.full(7, 1.0)
    return np.array([np.sum(wts * pts[:, 0] ** p * pts[:, 1] ** q) - moment(p, q) for p, q in MONOS])


def canonical(par):
    w1, a1, w2, a2, w3, b3, c3 = par
    o3 = sorted([(a1, w1), (a2, w2)])
    tri = sorted((1 - b3 - c3, b3, c3))
    return np.array([o3[0][0], o3[0][1], o3[1][0], o3[1][1], tri[0], tri[1], tri[2], w3])


def admissible(par, tol=1e-12):
    w1, a1, w2, a2, w3, b3, c3 = par
    a3 = 1 - b3 - c3
    inside = all(tol < a < 0.5 - tol for a in (a1, a2)) and min(a3, b3, c3) > tol
    distinct = abs(a1 - a2) > 1e-9 and min(abs(a3 - b3), abs(a3 - c3), abs(b3 - c3)) > 1e-9 and abs(a1 - 1 / 3) > 1e-9 and abs(a2 - 1 / 3) > 1e-9
    return inside and distinct and min(w1, w2, w3) > tol


def search(n_starts, seed=0):
    rng = np.random.default_rng(seed)
    roots, n_conv, n_adm = [], 0, 0
    for _ in range(n_starts):
        a1, a2 = rng.uniform(0.0, 0.5, 2)
        bc = rng.dirichlet((1.0, 1.0, 1.0))
        w = rng.uniform(0.0, 1.0 / 6.0, 3)
        x0 = np.array([w[0], a1, w[1], a2, w[2], bc[0], bc[1]])
        sol = root(residual, x0, method="hybr", tol=1e-15)
        if not sol.success or np.max(np.abs(residual(sol.x))) > 1e-13:
            continue
        n_conv += 1
        if not admissible(sol.x):
            continue
        pts, wts = rule(sol.x)
        worst = max(abs(np.sum(wts * pts[:, 0] ** p * pts[:, 1] ** q) - moment(p, q)) for p in range(7) for q in range(7 - p))
        if worst > 1e-13:
            continue
        n_adm += 1
        c = canonical(sol.x)
        if not any(np.max(np.abs(c - r)) < 1e-9 for r in roots):
            roots.append(c)
    return roots, n_conv, n_adm


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    roots, n_conv, n_adm = search(n)
    tab = json.loads((pathlib.Path(__file__).resolve().parents[1] / "proximalgalerkin_amd" / "tables" / "quadrature.json").read_text())["tri_deg6_12"]
    P, W = np.array(tab["points"]), np.array(tab["weights"])
    # the table in the same canonical form: orbits by weight multiplicity
    bary = np.column_stack([P, 1 - P.sum(axis=1)])
    groups = {}
    for b, w in zip(bary, W):
        groups.setdefault(round(w, 14), []).append(np.sort(b))
    o3 = sorted((min(g[0]), w) for w, g in groups.items() if len(g) == 3)
    (w6, g6), = [(w, g) for w, g in groups.items() if len(g) == 6]
    table_c = np.array([o3[0][0], o3[0][1], o3[1][0], o3[1][1], *g6[0], w6])
    out = {"n_starts": n, "converged_roots": n_conv, "admissible_roots": n_adm, "distinct_admissible_rules": len(roots),
           "canonical_form": "a1, w1, a2, w2 (3-orbits (a, a, 1-2a), a1 < a2), sorted barycentric point of the 6-orbit, w3",
           "rules": [r.tolist() for r in roots], "table": table_c.tolist(),
           "max_abs_difference_to_table": [float(np.max(np.abs(r - table_c))) for r in roots],
           "statement": "every admissible root of the [3,3,6] degree-6 moment system found by the global search is the committed table"
                        if len(roots) == 1 and np.max(np.abs(roots[0] - table_c)) < 1e-12 else "MORE THAN ONE RULE FOUND - see `rules`"}
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main()
