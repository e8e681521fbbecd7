#!/usr/bin/env python3
"""Build proximalgalerkin_amd/libpgx_stamp.so: the library with cycle stamps (s_memtime) in the interior-tile path of
k_st_smoothR<16,3,POST> - the fp64 smoother (since round 4 the A/B path, tuning key PGX_MG_F32=0) - for tools/smoother_stamps.py.  The stamps serialise the wave at every stamp (a scalar memory read and a
global store by lane 0), so the instrumented kernel is slower than the shipped one; the SHARES of the phases are what it is for.
    python tools/make_stamp_build.py && PGX_LIB=$PWD/proximalgalerkin_amd/libpgx_stamp.so python tools/smoother_stamps.py"""
import pathlib
import re
import subprocess

ROOT = pathlib.Path(__file__).resolve().parents[1]
SRC = ROOT / "proximalgalerkin_amd" / "csrc"
# every piece of pgx_kernels.hip this tool patches: tests/test_cpu_golden_abi_host.py asserts that they all still occur
HEAD = re.compile(r"template <int TY, int K, bool POST[^>]*>\n__device__ __forceinline__ void st_smoothR_fast\(")
TAIL = "// boundary tiles (and every tile of a level without uniform stencils):"
ANCHORS = [
    "  const int gi = i0 + lane;\n",
    "  if (POST) {\n    double xa[R], xc[R];",
    "    if (lj < H0 - 1) exch[lj * W + lane] = make_double2(rd[k][3], rd[k][5]);\n  }\n  __syncthreads();\n",
    "#pragma unroll\n  for (int s = 1; s <= K; ++s) {\n    const double2* const src",
    "    if (s < K) __syncthreads();\n  }\n}",
]


def missing_anchors():
    """Anchors that no longer occur in the function this tool instruments ([] = the tool can run)."""
    text = (SRC / "pgx_kernels.hip").read_text()
    m = HEAD.search(text)
    if not m or TAIL not in text[m.start():]:
        return ["st_smoothR_fast head / tail"]
    body = text[m.start(): text.index(TAIL, m.start())]
    return [a for a in ANCHORS if a not in body]


def sub(text, old, new):
    assert text.count(old) >= 1, old
    return text.replace(old, new, 1)


def main():
    assert not missing_anchors(), missing_anchors()
    s = (SRC / "pgx_kernels.hip").read_text()
    a = HEAD.search(s).start()
    b = s.index(TAIL, a)
    f = s[a:b]
    helper = '''#define PGX_NSTAMP 12
    __device__ unsigned long long g_stamps[8192][8][PGX_NSTAMP];
    __device__ __forceinline__ void stamp(int b, int wave, int i) {
      if ((threadIdx.x & 63) == 0 && b < 8192) g_stamps[b][wave][i] = __builtin_readcyclecounter();
    }
    extern "C" int pgx_debug_read_stamps(unsigned long long* out) {
      return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8192 * 8 * PGX_NSTAMP);
    }
    '''


    f = sub(f, "  const int gi = i0 + lane;\n", "  const int gi = i0 + lane;\n  if (POST && TY == 16) stamp(b, wave, 0);\n")
    f = sub(f, "  if (POST) {\n    double xa[R], xc[R];", "  if (POST && TY == 16) stamp(b, wave, 1);\n  if (POST) {\n    double xa[R], xc[R];")
    f = sub(f, "    if (lj < H0 - 1) exch[lj * W + lane] = make_double2(rd[k][3], rd[k][5]);\n  }\n  __syncthreads();\n",
            "    if (lj < H0 - 1) exch[lj * W + lane] = make_double2(rd[k][3], rd[k][5]);\n  }\n  if (POST && TY == 16) stamp(b, wave, 2);\n"
            "  __syncthreads();\n  if (POST && TY == 16) stamp(b, wave, 3);\n")
    f = sub(f, "#pragma unroll\n  for (int s = 1; s <= K; ++s) {\n    const double2* const src",
            "  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  if (POST && TY == 16) stamp(b, wave, 4);\n#pragma unroll\n"
            "  for (int s = 1; s <= K; ++s) {\n    const double2* const src")
    f = sub(f, "    if (s < K) __syncthreads();\n  }\n}",
            "    if (POST && TY == 16) stamp(b, wave, 3 + 2 * s);\n    if (s < K) __syncthreads();\n    if (POST && TY == 16) stamp(b, wave, 4 + 2 * s);\n  }\n}")
    tmp = SRC / "pgx_kernels_stamp_tmp.hip"
    tmp.write_text(s[:a] + helper + f + s[b:])
    try:
        flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value".split()
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", str(tmp), "-o", "/tmp/pgx_kernels_stamp.o"], cwd=SRC)
    finally:
        tmp.unlink()
    subprocess.check_call(["make"], cwd=SRC)
    objs = [str(SRC / (n + ".o")) for n in ("pgx_mg32", "pgx_p2", "pgx_patch", "pgx_comm", "pgx_nd", "pgx_gc", "pgx_sg", "pgx_qvi", "pgx_api")]
    out = ROOT / "proximalgalerkin_amd" / "libpgx_stamp.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "--offload-arch=gfx950", "-o", str(out), "/tmp/pgx_kernels_stamp.o", *objs, "-ldl", "-lpthread"])
    print(out)


if __name__ == "__main__":
    main()
