"""mgn_set_params alone, and inside a training loop (set_params + step! per iteration): python tools/set_params_time.py"""
import sys, time; sys.path.insert(0, ".")
import torch, numpy as np, mgn_amd, bench
ps = bench.glorot_params()
for dt in ("f32", "bf16"):
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15, dtype=dt)
    eng.set_params(ps)
    ts = []
    for _ in range(10):
        t = time.perf_counter(); eng.set_params(ps); ts.append(time.perf_counter() - t)
    print(dt, "mgn_set_params median %.2f ms" % (1e3 * sorted(ts)[5]), flush=True)
# a training loop as the reference drives it: new parameters before every step! (src/MeshGraphNets.jl:375-377 updates ps, the shim then calls
# mgn_set_params); the training kernels pack their own weights, the inference layouts are not touched
pos, cells, node_type, _ = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
N, E = pos.shape[0], s.size
rng = np.random.default_rng(0)
nf = rng.standard_normal((N, 9)).astype(np.float32); ef = rng.standard_normal((E, 3)).astype(np.float32)
tgt = rng.standard_normal((N, 2)).astype(np.float32); mask = np.arange(0, N, 2, dtype=np.int32)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(ps); eng.set_graph(s, r, N)
eng.step(nf, ef, tgt, mask)
ts = []
p2 = ps.copy()
for it in range(20):
    p2 *= 1.0001
    t = time.perf_counter(); eng.set_params(p2); gs, loss = eng.step(nf, ef, tgt, mask); ts.append(time.perf_counter() - t)
print("set_params + step! on the cylinder mesh: median %.2f ms" % (1e3 * sorted(ts)[10]))
t = time.perf_counter(); out = eng.forward(nf, ef); print("first forward after that (packs the inference layouts): %.2f ms" % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); out = eng.forward(nf, ef); print("second forward: %.2f ms" % (1e3 * (time.perf_counter() - t)))
