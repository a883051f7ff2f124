"""Wall time of mgn_forward (host in/out) on the cylinder mesh and its device-time split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgn_amd, bench
pos, cells, ntype, vel = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
N, E = pos.shape[0], s.size
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, N)
rng = np.random.default_rng(0)
nf = rng.standard_normal((N, 9)).astype(np.float32); ef = rng.standard_normal((E, 3)).astype(np.float32)
for _ in range(3): eng.forward(nf, ef)
t = time.perf_counter()
for _ in range(50): eng.forward(nf, ef)
print("mgn_forward wall, hipGraph replay: %.0f us" % ((time.perf_counter() - t) / 50 * 1e6))
eng.profile_enable(True)
t = time.perf_counter()
for _ in range(20): eng.forward(nf, ef)
dt = (time.perf_counter() - t) / 20
p = eng.profile_read()
print("mgn_forward wall with per-kernel events (eager): %.0f us" % (dt * 1e6))
for k, v in p.items():
    if v["count"]: print("  %-14s %6.1f us x %d" % (k, v["avg_ms"] * 1e3, v["count"] // 20))
