"""Check of mgn_step at a mid size (default 300x300 grid: N = 90 000, E = 537 602, 2 steps): many weight-gradient blocks,
rows_per_block > 32; loss and gradients against the float64 oracle, and the time per step.
usage: step_midsize.py [grid_side [mps]]   (grid_side 100: 1 870 edge tiles, still the cooperative / overlapped regime)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, mgn_amd, mgn_oracle as orc
side = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=2, mps=mps)
pos, s, r = mgn_amd.synth.mesh_1m(1234, side, side)
N, E = pos.shape[0], s.size
ps = orc.init_params(9, 3, 2, 128, 2, mps, 1234, 0.05)
rng = np.random.default_rng(0)
nf = rng.standard_normal((N, 9)).astype(np.float32); ef = rng.standard_normal((E, 3)).astype(np.float32)
tgt = rng.standard_normal((N, 2)).astype(np.float32); mask = np.arange(0, N, 3, dtype=np.int32)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, mps); eng.set_params(ps); eng.set_graph(s, r, N)
gs, loss = eng.step(nf, ef, tgt, mask)
ts = []
for _ in range(5):
    t = time.time(); gs, loss = eng.step(nf, ef, tgt, mask); ts.append(time.time() - t)
dt = min(ts)
t = time.time(); ref, rl = orc.step_grads(ps, cfg, nf, ef, s, r, tgt, mask); to = time.time() - t
print("N", N, "E", E, "step %.1f ms, oracle %.1f s" % (dt * 1e3, to))
print("loss", loss, rl, "grad rel L2", float(np.linalg.norm(gs - ref) / np.linalg.norm(ref)), "max rel", float(np.abs(gs - ref).max() / np.abs(ref).max()))
