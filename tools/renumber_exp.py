"""Experiment: does the node numbering (L2 locality of the P/Q gathers) matter for the edge kernel on M-1M?
Runs the processor bench with the generator's row-major numbering, a Morton (Z-curve) numbering, 32x32-blocked numbering
and a random numbering of the same mesh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgn_amd, bench

def part1by1(x):
    x = x.astype(np.uint64) & 0xFFFF
    x = (x | (x << 8)) & 0x00FF00FF
    x = (x | (x << 4)) & 0x0F0F0F0F
    x = (x | (x << 2)) & 0x33333333
    x = (x | (x << 1)) & 0x55555555
    return x

nx = 1000
pos, s, r = mgn_amd.synth.mesh_1m(1234, nx, nx)
N = pos.shape[0]
ix, iy = np.arange(N) % nx, np.arange(N) // nx
orders = {
    "row-major": np.arange(N),
    "morton": np.argsort(part1by1(ix) | (part1by1(iy) << 1), kind="stable"),
    "blocked32": np.argsort((iy // 32) * (nx // 32 + 1) * 1024 + (ix // 32) * 1024 + (iy % 32) * 32 + ix % 32, kind="stable"),
    "blocked8x128": np.argsort((iy // 8) * (nx // 128 + 1) * 1024 + (ix // 128) * 1024 + (iy % 8) * 128 + ix % 128, kind="stable"),
    "random": np.random.default_rng(0).permutation(N),
}
ps = bench.glorot_params()
for name, order in orders.items():
    new_id = np.empty(N, np.int32); new_id[order] = np.arange(N, dtype=np.int32)
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
    eng.set_params(ps); eng.set_graph(new_id[s], new_id[r], N); eng.latents_randn(1)
    eng.processor_steps_dev(15); eng.synchronize()
    eng.profile_enable(True)
    t = time.perf_counter()
    for _ in range(3): eng.processor_steps_dev(15)
    eng.synchronize(); dt = (time.perf_counter() - t) / 45
    p = eng.profile_read()
    print("%-12s step %.3f ms  edge %.3f ms  node %.3f ms" % (name, dt * 1e3, p["edge_step"]["avg_ms"], p["node_step"]["avg_ms"]), flush=True)
    eng.close()
