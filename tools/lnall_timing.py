"""Cost of the whole-array LayerNorm mode (mgn_config.ln_dims = MGN_LN_ALL, unfused driver) beside the default on the cylinder mesh:
right-hand side (resident inputs), 100-save Euler rollout, training step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mgn_amd, bench
pos, cells, ntype, vel = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
N, E = pos.shape[0], s.size
onehot = np.eye(7, dtype=np.float32)[ntype]
rel = pos[s] - pos[r]
ef = np.concatenate([rel, np.linalg.norm(rel, axis=1, keepdims=True)], 1).astype(np.float32)
rng = np.random.default_rng(0)
nf = rng.standard_normal((N, 9)).astype(np.float32); tgt = rng.standard_normal((N, 2)).astype(np.float32)
mask = np.nonzero(np.isin(ntype, [0, 5]))[0].astype(np.int32)
for dims in ("rows", "all"):
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15, ln_dims=dims)
    eng.set_params(bench.glorot_params()); eng.set_graph(s, r, N)
    eng.set_static(onehot, ef, np.ones(N, np.float32))
    for _ in range(4): eng.ode_step(vel)
    t = time.perf_counter()
    for _ in range(50): eng.ode_step(vel)
    rhs = (time.perf_counter() - t) / 50
    eng.rollout("Euler", vel, onehot, ef, 0.0, 1.0, 0.01, 101, dt=0.01)
    t = time.perf_counter(); eng.rollout("Euler", vel, onehot, ef, 0.0, 1.0, 0.01, 101, dt=0.01); ro = time.perf_counter() - t
    for _ in range(4): eng.step(nf, ef, tgt, mask)
    ts = []
    for _ in range(20):
        t = time.perf_counter(); eng.step(nf, ef, tgt, mask); ts.append(time.perf_counter() - t)
    print("ln_dims = %-4s: right-hand side %.0f us, Euler 100 saves %.1f ms, mgn_step %.2f ms" % (dims, rhs * 1e6, ro * 1e3, float(np.median(ts)) * 1e3))
    eng.close()
