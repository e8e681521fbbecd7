import sys
sys.path.insert(0, "/root/repo")
from proximalgalerkin_amd import fem
from proximalgalerkin_amd.gradient_constraint import GradientConstraintProblem, f_default, phi_default
N = int(sys.argv[1])
p = GradientConstraintProblem(fem.create_unit_square(N, N), phi_default, f_default)
p._opts.monitor = 2
for i in range(4):
    p.set_alpha(2.0**i)
    print("step", i + 1, p.solve(), flush=True)
    p.advance_prev()
