"""CPU oracle run behind tests/golden/obstacle_p2_n256_settingsB_divergence.json: example 01, P2, N = 256, settings B.
The exact-Newton (SuperLU) oracle itself ends with SNES_DIVERGED_DTOL at the alpha 16 -> 85 step - undamped Newton
(`snes_linesearch_type none`, the reference's setting) overshoots; the HIP path reproduces that outcome.
python tools/p2_divergence_oracle.py 256 > log   (about 100 CPU-minutes)"""
import sys, time
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
import numpy as np
from oracle import pg_oracle as O
N = int(sys.argv[1])
coords, cells = O.create_rectangle(N, N)
prob = O.ObstacleLagrange(coords, cells, degree=2)
log = O.NewtonLog()
t = time.time()
try:
    x, h = O.solve_problem(prob, 500, "double_exponential", 1e2, 1e-4, verbose=True, log=log)
    print("newton", h["Newton steps"], "time", time.time() - t)
except Exception as e:
    print("EXC", e)
print(log.fnorms)
