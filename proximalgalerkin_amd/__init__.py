"""proximalgalerkin_amd - MI355X-native LVPP / proximal-Galerkin Newton inner loop.

Host-side mirror of the reference's solver surface (lvpp.SNESProblem / SNESSolver,
dolfinx NonlinearProblem call shape) over the C ABI of include/pgx.h -> hand-written HIP (gfx950).
"""
from . import fem, ufl
from .fem import (Constant, Function, Mesh, QuadratureFunction, create_disk, create_rectangle, create_unit_square, dirichletbc,
                  functionspace)
from .problem import (ConvergenceError, NonlinearProblem, ObstacleResidual, SNESProblem, SNESSolver, derivative)

__all__ = [
    "SNESProblem", "SNESSolver", "NonlinearProblem", "ObstacleResidual", "derivative", "ConvergenceError", "fem", "ufl",
    "Mesh", "Function", "Constant", "QuadratureFunction", "create_disk", "create_rectangle", "create_unit_square", "dirichletbc",
    "functionspace",
]
