"""Time mgn_step (step!) on a cylinder_flow-sized datapoint and report the worst gradient error vs the float64 oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch, mgn_amd
import mgn_oracle as orc

cfg = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=2, mps=15)
pos, cells, node_type, vel = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
N, E = pos.shape[0], s.size
ps = orc.init_params(9, 3, 2, 128, 2, 15, 1234, 0.05)
rng = np.random.default_rng(0)
nf = rng.standard_normal((N, 9)).astype(np.float32); ef = rng.standard_normal((E, 3)).astype(np.float32)
tgt = rng.standard_normal((N, 2)).astype(np.float32)
mask = np.nonzero(np.isin(node_type, [0, 5]))[0].astype(np.int32)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(ps); eng.set_graph(s, r, N)
gs, loss = eng.step(nf, ef, tgt, mask)
K = 40
def timed(fn):
    ts = []
    for _ in range(K):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return float(np.median(ts))
dt = timed(lambda: eng.step(nf, ef, tgt, mask))                      # fresh host gradient vector per call
buf = np.zeros(eng.param_count, np.float32)
dt_reuse = timed(lambda: eng.step(nf, ef, tgt, mask, out=buf))       # caller-owned host vector
d = lambda a: torch.from_numpy(a).cuda()
nf_d, ef_d, tgt_d, gs_d = d(nf), d(ef), d(tgt), torch.zeros(eng.param_count, device="cuda")
eng.step(nf_d, ef_d, tgt_d, mask, out=gs_d)
dt_dev = timed(lambda: eng.step(nf_d, ef_d, tgt_d, mask, out=gs_d))  # graph and gradients stay on the device
assert np.array_equal(gs_d.cpu().numpy(), buf)
print("mgn_step median of %d: %.2f ms (fresh host gs)  %.2f ms (reused host gs)  %.2f ms (device in / out)" % (K, dt * 1e3, dt_reuse * 1e3, dt_dev * 1e3))
t = time.time(); out = eng.forward(nf, ef); tf = time.time() - t
print("N", N, "E", E, "mgn_step %.2f ms  (mgn_forward %.2f ms)" % (dt * 1e3, tf * 1e3))
if "--check" in sys.argv:
    t = time.time(); ref, rl = orc.step_grads(ps, cfg, nf, ef, s, r, tgt, mask); print("oracle %.1f s" % (time.time() - t))
    print("loss", loss, rl, "grad max rel (global)", float(np.abs(gs - ref).max() / np.abs(ref).max()))
