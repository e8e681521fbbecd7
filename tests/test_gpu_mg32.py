"""The single-precision V-cycle (csrc/pgx_mg32.hip, round 4) in every configuration of its template space that the default path does
not reach: two sweeps per launch (mg_nu = 4), one launch per leg (mg_nu = 3 / 2: the first launch also restricts), six-sweep
launches off, the residual + restriction as a launch of its own, single precision on the top levels only (fp64 right-hand side
down, fp64 correction up in mid-hierarchy), the un-normalised Krylov basis off, stream-synchronising read-backs - each against the
fp64 cycle of round 3 (PGX_MG_F32=0) AND against the CPU oracle: identical Newton counts, primal field within 1e-10.  The cycle is a
preconditioner inside FGMRES, so none of these may change what a Newton step computes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DOMAIN = ((-1.0, -1.0), (1.0, 1.0))
BASE = {"snes_linesearch_type": "none", "snes_rtol": 1e-6, "snes_max_it": 100, "snes_error_if_not_converged": True}


def _solve(N, tuning, opts):
    from proximalgalerkin_amd import _lib, fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    for k, v in tuning.items():
        _lib.tuning_set(k, v)
    try:
        msh = fem.create_rectangle(DOMAIN, (N, N))
        problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=dict(BASE, **opts))
        hist = run_outer_loop(problem, sol, sol_k, alpha, 100, "double_exponential", 1e2, 1e-4, verbose=False)
        x = sol.x.array.copy()
        problem.close()
    finally:
        for k in tuning:
            _lib.tuning_set(k, None)
    return x, hist


@pytest.fixture(scope="module")
def reference(require_gpu):
    """The exact-Newton SuperLU oracle's run at 256^2 (tests/golden/, tools/make_golden_large.py): five single-precision levels above
    the fused tail, tiles of every kind (interior, frame) on the top ones.  The fp64 cycle of round 3 must reproduce it too."""
    import pathlib

    g = np.load(pathlib.Path(__file__).parent / "golden" / "obstacle_p1_n256_settingsB_large.npz")
    N, u_ref, counts = int(g["N"]), g["u_final"], [int(c) for c in g["hist_Newton_steps"]]
    x64, h64 = _solve(N, {"PGX_MG_F32": 0}, {})
    assert h64["Newton steps"] == counts
    assert np.linalg.norm(x64[: len(u_ref)] - u_ref) <= 1e-10 * np.linalg.norm(u_ref)
    return N, len(u_ref), u_ref, {"Newton steps": counts}


@pytest.mark.parametrize("tuning,opts", [
    ({}, {}),                                                    # default: K = 3 on the big levels, six-sweep launches below 513^2
    ({}, {"mg_nu": 4}),                                          # two sweeps per launch, two launches per leg
    ({}, {"mg_nu": 3}),                                          # one launch per leg: the FIRST launch restricts (levels <= 513^2)
    ({"PGX_F32_RR_MAX": 0}, {"mg_nu": 2}),                       # one two-sweep launch per leg, restriction as its own launch
    ({"PGX_F32_K6_MAX": 0}, {}),                                 # three sweeps per launch everywhere, fused restriction
    ({"PGX_F32_K6_MAX": 0, "PGX_F32_RR_MAX": 0}, {}),            # ... and the restriction as its own launch
    ({"PGX_F32_MIN": 20000}, {}),                                # single precision on 257^2 and 129^2... only: fp64 levels in between
    ({"PGX_F32_TY": 16}, {}),                                    # 16-row tiles on every level
    ({"PGX_F32_TY": 8, "PGX_F32_K6_MAX": 0}, {}),                # 8-row tiles
    ({"PGX_LAZY_NORM": 0}, {}),                                  # Krylov vectors normalised in place
    ({"PGX_HOST_POLL": 0}, {}),                                  # read-backs through hipMemcpyAsync + hipStreamSynchronize
    ({"PGX_SPMV_D4": 0}, {}),                                    # operator apply from the four D arrays
    ({"PGX_MG_MIN_NX": 2}, {}),                                  # 3 x 3 coarsest grid (round 3)
    ({"PGX_Z_F32": 0}, {}),                                      # FGMRES Z_j as fp64 pairs (round 4) instead of one float2 field
])
def test_single_precision_cycle_variants_match_the_oracle(reference, tuning, opts):
    N, n, x_ref, h_ref = reference
    x, h = _solve(N, tuning, opts)
    assert h["Newton steps"] == h_ref["Newton steps"], (tuning, opts)
    assert np.linalg.norm(x[:n] - x_ref[:n]) <= 1e-10 * np.linalg.norm(x_ref[:n]), (tuning, opts)


def _solve_rect(cells, tuning, opts, domain):
    from proximalgalerkin_amd import _lib, fem
    from proximalgalerkin_amd.obstacle import run_outer_loop, setup_problem

    for k, v in tuning.items():
        _lib.tuning_set(k, v)
    try:
        msh = fem.create_rectangle(domain, cells)
        problem, sol, sol_k, alpha = setup_problem(msh, 1, petsc_options=dict(BASE, **opts))
        hist = run_outer_loop(problem, sol, sol_k, alpha, 100, "double_exponential", 1e2, 1e-4, verbose=False)
        x = sol.x.array.copy()
        problem.close()
    finally:
        for k in tuning:
            _lib.tuning_set(k, None)
    return x, hist


@pytest.mark.parametrize("cells", [(200, 72), (130, 260), (96, 48), (330, 118), (90, 270), (258, 130), (124, 124)])
def test_single_precision_cycle_on_rectangular_grids_that_cut_every_tile(require_gpu, cells):
    """Grids whose sides are no multiples of the tile sizes (58 / 52 / 48 columns, 8 / 16 rows), long and thin in either direction,
    with odd coarse levels: the boundary sub-tiles, the last partial tiles and the coarse-correction reads of every launch flavour
    on shapes the square goldens do not have.  Single precision is forced down to levels of 500 vertices (PGX_F32_MIN) so that two
    to four levels run it.  Against the fp64 cycle on the same mesh: identical Newton counts, primal field within 1e-10; and the
    default path (single precision above 4 000 vertices) as well."""
    domain = ((-1.0, -1.0), (1.0, 1.0))  # cells up to 3 : 1 (a point smoother is not made for the 16 : 1 cells of, say, 640 x 40)
    x64, h64 = _solve_rect(cells, {"PGX_MG_F32": 0}, {}, domain)
    n = (cells[0] + 1) * (cells[1] + 1)
    for tuning in ({"PGX_F32_MIN": 500}, {}, {"PGX_F32_MIN": 500, "PGX_F32_K6_MAX": 0, "PGX_F32_RR_MAX": 0}):
        x, h = _solve_rect(cells, tuning, {}, domain)
        assert h["Newton steps"] == h64["Newton steps"], (cells, tuning)
        assert np.linalg.norm(x[:n] - x64[:n]) <= 1e-10 * np.linalg.norm(x64[:n]), (cells, tuning)


def test_float_storage_of_the_krylov_z_vectors_changes_nothing(require_gpu):
    """Round 5: the Z_j of FGMRES are outputs of the single-precision cycle, stored as ONE float2 field each (the cycle's last launch
    writes them, k_st_spmv_r<true> and k_lincomb_f2 read them).  A float cast to double and back is the identity, so against the
    fp64-pair storage (PGX_Z_F32=0) every Newton count must agree and the final iterate must be BITWISE the same."""
    a, ha = _solve(128, {}, {})
    b, hb = _solve(128, {"PGX_Z_F32": 0}, {})
    assert ha["Newton steps"] == hb["Newton steps"]
    assert np.array_equal(a, b)
