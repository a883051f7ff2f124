"""What the whole-array LayerNorm mode (mgn_config.ln_dims = MGN_LN_ALL) costs per processor step: the unfused path (MLP kernel with its
LayerNorm off, grid-wide statistics, apply pass) against the fused default, on an nx x nx slice of the M-1M generator.
python tools/ln_all_cost.py [nx]   (host arrays in / out: the difference of an n-step and a 0-step call is the device time)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np, mgn_amd, bench
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pos, s, r = mgn_amd.synth.mesh_1m(1234, nx, nx)
N, E = pos.shape[0], s.size
rng = np.random.default_rng(0)
v = rng.standard_normal((N, 128), dtype=np.float32); e = rng.standard_normal((E, 128), dtype=np.float32)
ps = bench.glorot_params()
for name, kw in (("ln_dims=all (unfused)", dict(ln_dims="all")), ("default (fused, split path)", dict())):
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15, **kw)
    eng.set_params(ps); eng.set_graph(s, r, N)
    eng.processor_steps(v, e, 1)
    ts = {}
    for n in (0, 4):
        t0 = time.perf_counter(); eng.processor_steps(v, e, n); ts[n] = time.perf_counter() - t0
    print(f"{name:30s} N={N} E={E}: {(ts[4] - ts[0]) / 4 * 1e3:8.2f} ms per processor step (4-step call {ts[4]:.2f} s, 0-step call {ts[0]:.2f} s)")
    eng.close()
