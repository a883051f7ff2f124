import sys, time; sys.path.insert(0, ".")
import torch, numpy as np, mgn_amd, bench
ps = bench.glorot_params()
pos, cells, _, _ = mgn_amd.synth.mesh_cyl(1, 100)
s, r = mgn_amd.synth.cells_to_edges(cells)
N, E = pos.shape[0], s.size
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(ps); eng.set_graph(s, r, N)
rng = np.random.default_rng(0)
nf = rng.standard_normal((N, 9)).astype(np.float32); ef = rng.standard_normal((E, 3)).astype(np.float32)
for _ in range(3): eng.forward(nf, ef)
ts = []
for it in range(8):
    eng.set_params(ps * np.float32(1 + 1e-4 * it))
    t = time.perf_counter(); eng.forward(nf, ef); t1 = time.perf_counter() - t
    t = time.perf_counter(); eng.forward(nf, ef); t2 = time.perf_counter() - t
    t = time.perf_counter(); eng.forward(nf, ef); t3 = time.perf_counter() - t
    ts.append((t1, t2, t3))
print("forward after set_params / 2nd / 3rd (ms):", [tuple(round(x * 1e3, 2) for x in t) for t in ts[2:6]])
