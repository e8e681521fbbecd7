"""One full LVPP solve of example 02 (reference defaults: E 2e4, nu 0.3, disp -0.25, gap 0, doubling alpha, tol 1e-6) on
an n x n x n cube of 6 n^3 tetrahedra: python tools/sg_scaling.py n"""
import sys
import time

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from proximalgalerkin_amd import signorini as G  # noqa: E402

import os  # noqa: E402

n = int(sys.argv[1])
mesh = G.create_unit_cube(n, n, n)
mt, bcs = G.native_tags(mesh)
comm, device = None, 0
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 tools/sg_scaling.py 70
    # one process per GPU, distributed sparse LU over RCCL (pgx_sg_create_dist); NOT executed on this pool's one-GPU
    # boxes - the same code path runs in tests/test_gpu_nd_dist.py through the in-process transport
    import torch.distributed as dist

    from proximalgalerkin_amd import comm as pcomm

    device = int(os.environ.get("LOCAL_RANK", "0"))
    dist.init_process_group("gloo")
    comm = pcomm.rccl_from_torch_distributed(device)
t = time.perf_counter()
problem = G.SignoriniProblem(mesh, mt.find(2), np.unique(mt.find(1).ravel()), 2.0e4, 0.3, 0.0, -0.25, device=device, comm=comm)
print(f"n={n} tets={mesh.cells.shape[0]} dofs u={3 * problem.nv} psi={problem.npsi} setup {time.perf_counter() - t:.2f}s", flush=True)
problem.profile(True)
t = time.perf_counter()
its = []
for it in range(1, 26):
    problem.set_alpha(2.0**it)
    tol = 1e-5 if it < 2 else 1e-6
    problem.solver.setTolerances(atol=tol, rtol=tol)
    t1 = time.perf_counter()
    reason, k = problem.solve()
    d = problem.u_increment()
    its.append(k)
    print(f"  it {it}: alpha={2.0**it:g} reason={reason} newton={k} |du|={d:.3e} ({(time.perf_counter() - t1) * 1e3:.0f} ms)", flush=True)
    if d <= 1e-6:
        break
    problem.advance_prev()
dt = time.perf_counter() - t
print(f"  total {dt:.2f}s, Newton {its} sum {sum(its)} -> {sum(its) / dt:.2f} Newton it/s", flush=True)
print(f"  sparse LU storage {problem.lu_stats()['arena_doubles'] * 8 / 1e9:.1f} GB", flush=True)
print("  phases ms:", {k: round(v, 1) for k, v in problem.profile(False).items()}, flush=True)
