"""Per-family kernel times of one full forward (encode + 15 steps + decode) on M-1M, device side only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgn_amd, bench
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
pos, s, r = mgn_amd.synth.mesh_1m(1234, nx, nx)
N, E = pos.shape[0], s.size
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15, dtype=dtype)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, N)
rng = np.random.default_rng(0)
nf = rng.standard_normal((N, 9)).astype(np.float32); ef = rng.standard_normal((E, 3)).astype(np.float32)
eng.fwd_upload(nf, ef)
for it in range(3):
    if it == 1: eng.profile_enable(True)
    eng.fwd_encode()
    for k in range(15):
        eng.proc_edge(k); eng.proc_node(k, k < 14)
    eng.fwd_decode()
eng.synchronize()
p = eng.profile_read()
tot = sum(v["avg_ms"] * v["count"] for v in p.values()) / 2
print({k: (round(v["avg_ms"], 3), v["count"]) for k, v in p.items()}, "device ms per forward", round(tot, 2))
t = time.time(); out = eng.forward(nf, ef); print("mgn_forward incl. H2D/D2H + host scatter: %.1f ms" % ((time.time() - t) * 1e3))
eng.set_norms(); onehot = nf[:, 2:].copy(); x = nf[:, :2].copy()
eng.set_static(onehot, ef)
t = time.time(); d = eng.ode_step(x); print("ode_step fast path: %.1f ms" % ((time.time() - t) * 1e3))
t = time.time(); d = eng.ode_step(x, onehot, ef); print("ode_step one-shot: %.1f ms" % ((time.time() - t) * 1e3))
