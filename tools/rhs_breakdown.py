"""Device time of one right-hand side (mgn_ode_step after mgn_set_static) on the cylinder mesh, by kernel family."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgn_amd, bench
pos, cells, ntype, vel = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
N = pos.shape[0]
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, N)
onehot = np.eye(7, dtype=np.float32)[ntype]
rel = pos[s] - pos[r]
ef = np.concatenate([rel, np.linalg.norm(rel, axis=1, keepdims=True)], 1).astype(np.float32)
eng.set_static(onehot, ef, np.ones(N, np.float32))
for _ in range(3): eng.ode_step(vel)
t = time.perf_counter()
for _ in range(50): out = eng.ode_step(vel)
print("wall per RHS, hipGraph replay (what a host-driven solver sees): %.0f us" % ((time.perf_counter() - t) / 50 * 1e6))
eng.profile_enable(True)
t = time.perf_counter()
for _ in range(20): eng.ode_step(vel)
dt = (time.perf_counter() - t) / 20
p = eng.profile_read()
print("wall per RHS with per-kernel events (eager): %.0f us" % (dt * 1e6))
for k, v in p.items():
    if v["count"]: print("  %-14s %6.1f us x %d per RHS" % (k, v["avg_ms"] * 1e3, v["count"] // 20))
