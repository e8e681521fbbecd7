"""Mid-size goldens of examples 06, 02 (degrees 1, 2) and 01-P2 (tools/make_golden_families.py: CPU oracle runs of minutes, too long to
repeat inside a GPU test): the HIP path reproduces the per-step Newton counts and the final primal field to 1e-10 relative L2 (ex 01-P2,
ex 06 up to 48^2) resp. to the accuracy the examples' own Newton tolerances define (see the tests) on
meshes where its sparse LU works on a deep dissection tree (several size classes per depth, subtree batches)."""
import pathlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = pathlib.Path(__file__).resolve().parent / "golden"


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _files(pattern):
    return sorted(GOLD.glob(pattern))


@pytest.mark.parametrize("f", _files("gradient_constraint_p2_n*_defaults_mid.npz"), ids=lambda f: f.stem)
def test_example06(require_gpu, f):
    from proximalgalerkin_amd.gradient_constraint import solve_problem

    z = np.load(f)
    N = int(z["N"])
    its, _, x = solve_problem(N, N, verbose=False, return_solution=True)
    assert list(its) == list(z["newton"]), (its, z["newton"])
    n2 = z["u_final"].size
    # 1e-10 up to 48^2; at 96^2 the two runs differ by 2.2e-10: the late Newton matrices have condition numbers beyond 1e10, the
    # oracle's SuperLU steps carry no iterative refinement (the HIP path refines every solve to a 1e-12 true residual), and SNES
    # stops both at rtol = atol = 1e-9 - the field is determined to a few 1e-10 by the problem's own tolerances
    assert _rel(x[:n2], z["u_final"]) < (1e-10 if N <= 48 else 5e-10)


@pytest.mark.parametrize("f", _files("signorini_p*_n*_defaults_mid.npz"), ids=lambda f: f.stem)
def test_example02(require_gpu, f):
    from proximalgalerkin_amd import signorini as G

    z = np.load(f)
    n, degree = int(z["n"]), int(z["degree"])
    mesh = G.create_unit_cube(n, n, n)
    mt, bcs = G.native_tags(mesh)
    it, iterations, x, cv = G.solve_contact_problem(mesh, mt, bcs, degree=degree, verbose=False, return_solution=True)
    assert it == int(z["it"]) and list(iterations) == list(z["newton"])
    nu3 = z["u_final"].size
    # measured: 14^3 and 8^3 (degree 2) agree to < 1e-10, 20^3 to 4.1e-10, 12^3 (degree 2) to 6.5e-10.  The reference's Newton
    # tolerance for this example is 1e-6 (signorini_dolfinx.py:330-332): the last step of either implementation lands wherever
    # quadratic convergence takes it below that, and the oracle's SuperLU steps are not refined - 1e-9 is what the two share
    assert _rel(x[:nu3], z["u_final"]) < 1e-9


@pytest.mark.parametrize("f", _files("signorini_hex_q*_defaults_mid.npz"), ids=lambda f: f.stem)
def test_example02_on_the_references_native_hexahedral_mesh(require_gpu, f):
    """signorini_dolfinx.py with no arguments: create_unit_cube(16, 7, 5, hexahedron), degree 2."""
    from proximalgalerkin_amd import signorini as G

    z = np.load(f)
    mesh = G.create_unit_cube_hex(*[int(v) for v in z["n"]])
    mt, bcs = G.native_tags(mesh)
    it, iterations, x, cv = G.solve_contact_problem(mesh, mt, bcs, degree=int(z["degree"]), verbose=False, return_solution=True)
    assert it == int(z["it"]) and list(iterations) == list(z["newton"])
    nu3 = z["u_final"].size
    assert _rel(x[:nu3], z["u_final"]) < 1e-9  # Newton tolerance 1e-6, see test_example02


@pytest.mark.parametrize("f", _files("obstacle_p2_n*_settingsB_mid.npz"), ids=lambda f: f.stem)
def test_example01_p2(require_gpu, f):
    from proximalgalerkin_amd import fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    z = np.load(f)
    N = int(z["N"])
    msh = fem.create_rectangle(((-1.0, -1.0), (1.0, 1.0)), (N, N))
    problem, sol, sol_k, alpha = setup_problem(msh, 2)
    hist = run_outer_loop(problem, sol, sol_k, alpha, 500, "double_exponential", 1e2, 1e-4)
    assert list(hist["Newton steps"]) == list(z["newton"]), (hist["Newton steps"], z["newton"])
    n = z["u_final"].size
    assert _rel(sol.x.array[:n], z["u_final"]) < 1e-10
    problem.close()


# ---- goldens from the CPU oracles with the nested-dissection LU (tools/make_golden_nd_families.py): the sizes of BASELINE configs 4 and 5 ----
def _chunk_sums(v, chunk):
    m = -(-v.size // chunk) * chunk
    w = np.zeros(m)
    w[: v.size] = v
    return w.reshape(-1, chunk).sum(axis=1)


def _lattice(g, stride, d):
    M = g.shape[0]
    nb = (M - 1) // stride
    sample = g[(slice(None, None, stride),) * d]
    core = g[(slice(0, nb * stride),) * d]
    shp = []
    for _ in range(d):
        shp += [nb, stride]
    return sample, core.reshape(*shp, *g.shape[d:]).sum(axis=tuple(range(1, 2 * d, 2)))


def _close(a, b, tol):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) <= tol * np.linalg.norm(np.asarray(b))


@pytest.mark.parametrize("f", _files("gradient_constraint_p2_n*_defaults_nd.npz"), ids=lambda f: f.stem)
def test_example06_against_the_nd_oracle_fingerprint(require_gpu, f):
    """Example 06 at the sizes whose oracle run needs the nested-dissection LU (config 4's own 1024^2 among them): identical
    per-step Newton counts, and a fingerprint of the final primal field in which every dof takes part - sub-lattice of the vertex
    values, sums over the stride x stride vertex blocks, chunk sums of the edge-midpoint dofs, 2-norm, extrema.  Tolerance: 1e-10
    up to 48^2 (test_example06); beyond, what SNES rtol = atol = 1e-9 on condition numbers past 1e10 leaves defined - measured
    2.2e-10 at 96^2 - so 5e-10, as there.  (gradient_constraint_dolfinx.py:100-132,171-205)"""
    from proximalgalerkin_amd.gradient_constraint import solve_problem

    z = np.load(f)
    N, stride = int(z["N"]), int(z["stride"])
    its, _, x = solve_problem(N, N, verbose=False, return_solution=True)
    assert list(its) == list(z["newton"]), (list(its), list(z["newton"]))
    nv = (N + 1) ** 2
    n2 = nv + (3 * N * N + 2 * N)  # vertices + edges of the right-diagonal N x N mesh
    u = x[:n2]
    tol = 1e-10 if N <= 48 else 5e-10
    sample, blocksum = _lattice(u[:nv].reshape(N + 1, N + 1), stride, 2)
    assert _close(sample, z["u_sample"], tol)
    assert _close(blocksum, z["u_blocksum"], tol)
    assert _close(_chunk_sums(u[nv:], int(z["edge_chunk"])), z["u_edge_chunksum"], tol)
    assert abs(np.linalg.norm(u) - float(z["u_norm2"])) <= tol * float(z["u_norm2"])
    assert abs(np.linalg.norm(u[:nv]) - float(z["u_vertex_norm2"])) <= tol * float(z["u_vertex_norm2"])
    assert abs(u.max() - float(z["u_max"])) <= 10 * tol * max(abs(float(z["u_max"])), 1e-3)
    assert abs(u.min() - float(z["u_min"])) <= 10 * tol * max(abs(float(z["u_max"])), 1e-3)


@pytest.mark.parametrize("f", _files("signorini_p1_n*_defaults_nd.npz"), ids=lambda f: f.stem)
def test_example02_against_the_nd_oracle_fingerprint(require_gpu, f):
    """Example 02 (P1 tetrahedra) at sizes whose oracle run needs the nested-dissection LU (config 5's own 70^3 among them):
    identical proximal and Newton counts and the fingerprint of the displacement field (sub-lattice per component, block sums,
    chunk sums of the flat dof vector, norm, extrema) to 1e-9 - the reference's Newton tolerance for this example is 1e-6
    (signorini_dolfinx.py:330-332), see test_example02."""
    from proximalgalerkin_amd import signorini as G

    z = np.load(f)
    n, stride = int(z["n"]), int(z["stride"])
    mesh = G.create_unit_cube(n, n, n)
    mt, bcs = G.native_tags(mesh)
    it, iterations, x, cv = G.solve_contact_problem(mesh, mt, bcs, degree=1, verbose=False, return_solution=True)
    assert it == int(z["it"]) and list(iterations) == list(z["newton"]), (it, list(iterations), list(z["newton"]))
    M = n + 1
    nv = M ** 3
    u = x[: 3 * nv]
    comp = np.stack([u[k * nv:(k + 1) * nv].reshape(M, M, M) for k in range(3)], axis=-1) if int(z["blocked"]) else u.reshape(M, M, M, 3)
    sample, blocksum = _lattice(comp, stride, 3)
    tol = 1e-9
    assert _close(sample, z["u_sample"], tol)
    assert _close(blocksum, z["u_blocksum"], tol)
    assert _close(_chunk_sums(u, int(z["chunk"])), z["u_chunksum"], tol)
    assert abs(np.linalg.norm(u) - float(z["u_norm2"])) <= tol * float(z["u_norm2"])
    scale = max(abs(float(z["u_max"])), abs(float(z["u_min"])))
    assert abs(u.max() - float(z["u_max"])) <= 10 * tol * scale and abs(u.min() - float(z["u_min"])) <= 10 * tol * scale
