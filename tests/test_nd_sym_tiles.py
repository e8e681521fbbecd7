"""The tile enumeration of the sparse LU's symmetric GEMM launches (csrc/pgx_nd_gemm.h nd_sym_tiles / nd_sym_tile, round 5) through its
host-side test hooks - pure integer arithmetic, no GPU: every tile (tr, tc) of an nr x nc rectangle with tc <= tr + band exactly
once, none other, in groups of eight tile rows (the locality the XCD-chunked order relies on)."""
import ctypes as C
import itertools

import pytest

from proximalgalerkin_amd import _lib


@pytest.mark.parametrize("band", [0, 1])
def test_symmetric_tile_enumeration_is_a_bijection(band):
    lib = _lib.load()
    tr, tc = C.c_int(0), C.c_int(0)
    for nr, nc in itertools.product([1, 2, 3, 7, 8, 9, 15, 16, 17, 33, 119, 238], [1, 2, 4, 8, 9, 33, 119]):
        if nc > nr:
            continue
        n = lib.pgx_nd_sym_tile_count(nr, nc, band)
        want = {(r, c) for r in range(nr) for c in range(nc) if c <= r + band}
        got = []
        for t in range(n):
            lib.pgx_nd_sym_tile_at(t, nr, nc, band, C.byref(tr), C.byref(tc))
            got.append((tr.value, tc.value))
        assert len(got) == len(want) and set(got) == want, (nr, nc, band)
        groups = [r // 8 for r, _ in got]
        assert groups == sorted(groups)  # group after group: a run of consecutive tiles stays within few tile rows
    # the square Schur block of example 02's root level: 119 x 119 tiles of 128 -> 7140 launched instead of 14161
    assert lib.pgx_nd_sym_tile_count(119, 119, 0) == 119 * 120 // 2
