import sys; sys.path.insert(0, '.')
import numpy as np, bench, dcreg_amd
from dcreg_amd import scenes as h
for name in ("c1_fixture_7562", "c2_cylinder_100k", "c3_pk01_200k", "c4_corridor_1m"):
    W = bench.WORKLOADS[name]
    tgt, src = bench.make_pair(W["scene"], W["n"], 100)
    c = dcreg_amd.Context(0); c.set_target(tgt, W["radius"]); c.set_source(src)
    i = c.index_info(); print(name, "cell %.4f" % i.cell, "R_s %.4f" % (1.05 * W["radius"]), "dims", tuple(i.dims), "cells", i.n_cells, "pts/cell(all)", round(len(tgt) / i.n_cells, 2)); c.close()
tgt, src = h.scene_parkinglot()
c = dcreg_amd.Context(0); c.set_target(tgt, 0.5); c.set_source(src); i = c.index_info(); print("c3_reg cell %.4f" % i.cell, tuple(i.dims)); c.close()
