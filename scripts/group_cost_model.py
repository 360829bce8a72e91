"""Host replay: per-group search cost (candidates, rows per query) against the distance of the group from the body origin, for a bench
workload at its initial pose - which groups are the heavy ones (profiles/r03_ablation.md section 10).  CPU only;
usage: python scripts/group_cost_model.py [workload]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import emul, bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3_pk01_200k"
W = bench.WORKLOADS[wl]
tgt, src = bench.make_pair(W["scene"], W["n"], seed=100)
T = bench.initial_pose(W["scene"])
idx = emul.Index(tgt, W["radius"])
S = emul.Source(src)
print("cell", idx.cell, idx.dims)
r = emul.linearize(idx, S, T[:3, :3], T[:3, 3], radius=W["radius"], stats=True)
st = r["stats"].astype(np.float64)
p = S.sorted.astype(np.float64)
far2 = (p ** 2).sum(axis=1)
G = 4096
ng = len(p) // G
cost = st[:ng * G, 0].reshape(ng, G).mean(axis=1)       # candidates per query
rows = st[:ng * G, 3].reshape(ng, G).mean(axis=1)
f = np.sqrt(far2[:ng * G].reshape(ng, G).max(axis=1))
order = np.argsort(-f)
print("groups far-first: (far m, cand/query, rows/query)")
for g in order: print("%7.1f %7.1f %6.1f" % (f[g], cost[g], rows[g]))
print("corr(far, cand) = %.2f" % np.corrcoef(f, cost)[0, 1])
