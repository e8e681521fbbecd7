import sys, time
sys.path.insert(0, ".")
import numpy as np
from proximalgalerkin_amd import fem, _lib
from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default
t = time.perf_counter(); lib = _lib.load(); print(f"load {time.perf_counter()-t:.2f}")
t = time.perf_counter(); mesh = fem.create_unit_square(1024, 1024); print(f"mesh {time.perf_counter()-t:.2f}")
t = time.perf_counter(); U = fem.FunctionSpace(mesh, 2, 1); xd = U.dof_coordinates(); cd = U.cell_dofs(); print(f"space tables {time.perf_counter()-t:.2f}")
t = time.perf_counter(); p = GradientConstraintProblem(mesh, phi_default, f_default); print(f"GradientConstraintProblem total {time.perf_counter()-t:.2f}")
