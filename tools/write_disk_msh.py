import sys
sys.path.insert(0, '.')
from proximalgalerkin_amd import fem
m = fem.create_disk(0.05)
with open('gpurun_out/disk.msh', 'w') as f:
    f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % m.num_vertices)
    for i, (x, y) in enumerate(m.geometry):
        f.write("%d %.17g %.17g 0\n" % (i + 1, x, y))
    f.write("$EndNodes\n$Elements\n%d\n" % m.num_cells)
    for i, c in enumerate(m.cells):
        f.write("%d 2 2 1 1 %d %d %d\n" % (i + 1, c[0] + 1, c[1] + 1, c[2] + 1))
    f.write("$EndElements\n")
