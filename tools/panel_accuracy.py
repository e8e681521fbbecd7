"""Backward error of PLAIN sparse-LU solves (no refinement) on late example-06 Newton matrices, MFMA panel solves (PGX_ND_PANEL=0:
16 x 16 diagonal blocks applied through their inverses) against the scalar substitution kernel (PGX_ND_PANEL=1):
python tools/panel_accuracy.py [N=256]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proximalgalerkin_amd import fem  # noqa: E402
from proximalgalerkin_amd.direct import DirectSolver  # noqa: E402
from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mesh = fem.create_unit_square(N, N)
p = GradientConstraintProblem(mesh, phi_default, f_default)
mats = []
for i in range(25):
    p.set_alpha(2.0**i)
    r, n = p.solve()
    if i in (3, 8, 12):
        x = p.get_state()
        J = p.jacobian(x).tocsr()
        J.sort_indices()
        mats.append((i, J, np.abs(x[p.n2:]).max()))
    if p.l2_increment() < 1e-8:
        x = p.get_state()
        J = p.jacobian(x).tocsr()
        J.sort_indices()
        mats.append((i, J, np.abs(x[p.n2:]).max()))
        break
    p.advance_prev()
nod = np.concatenate([np.arange(p.n2), np.arange(p.nv), np.arange(p.nv)])
coords = p.U.dof_coordinates()
p.close()
rng = np.random.default_rng(0)
for i, J, psimax in mats:
    b = rng.standard_normal(J.shape[0])
    nrmJ = abs(J).sum(axis=0).max()
    for kind in (1, 0):
        os.environ["PGX_ND_PANEL"] = str(kind)
        ds = DirectSolver(J.indptr, J.indices, nod, coords, device=0)
        ds.factor(J.data)
        x = ds.solve(b)
        be = np.linalg.norm(J @ x - b) / (nrmJ * np.linalg.norm(x) + np.linalg.norm(b))
        x2 = x + ds.solve(b - J @ x)
        be2 = np.linalg.norm(J @ x2 - b) / (nrmJ * np.linalg.norm(x2) + np.linalg.norm(b))
        print(f"N={N} LVPP step {i:2d} alpha=2^{i} max|psi|={psimax:.2e} |J|_1={nrmJ:.2e} panel kind {kind}: backward error {be:.2e}, "
              f"after one refinement step {be2:.2e}, perturbed pivots {ds.stats()['perturbed_pivots']}", flush=True)
        ds.close()
