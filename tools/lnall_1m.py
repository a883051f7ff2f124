"""mgn_forward on M-1M under ln_dims = MGN_LN_ALL (unfused driver) beside the default: ms per forward, ms per processor step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, mgn_amd, bench
pos, s, r = mgn_amd.synth.mesh_1m(1234)
N, E = pos.shape[0], s.size
rng = np.random.default_rng(1)
nf = rng.standard_normal((N, 9), dtype=np.float32); ef = rng.standard_normal((E, 3), dtype=np.float32)
for dims in ("rows", "all"):
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15, ln_dims=dims)
    eng.set_params(bench.glorot_params()); eng.set_graph(s, r, N)
    eng.forward(nf, ef)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); out = eng.forward(nf, ef); ts.append(time.perf_counter() - t)
    print("ln_dims = %-4s: mgn_forward %.1f ms (host in / out included), %.2f ms per processor step at most" % (dims, min(ts) * 1e3, min(ts) * 1e3 / 15))
    eng.close()
