"""Soak: a training-like loop that keeps changing trajectory (mgn_set_graph with meshes of different sizes), parameters and
entry points; device memory in use must stop growing after the first rounds (buffers grow to the largest mesh and stay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np, mgn_amd, bench
import psutil
_proc = psutil.Process()

def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 1e6

ps = bench.glorot_params()
meshes = []
for seed, n in ((1, 1500), (2, 2600), (3, 900)):
    pos, cells, ntype, vel = mgn_amd.synth.mesh_cyl(seed, n)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    rel = pos[s] - pos[r]
    ef = np.concatenate([rel, np.linalg.norm(rel, axis=1, keepdims=True)], 1).astype(np.float32)
    meshes.append((pos, s, r, np.eye(7, dtype=np.float32)[ntype], ef, vel, ntype))
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
rng = np.random.default_rng(0)
log = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    pos, s, r, onehot, ef, vel, ntype = meshes[it % 3]
    N, E = pos.shape[0], s.size
    eng.set_params(ps * (1.0 + 1e-3 * (it % 5)))
    eng.set_graph(s, r, N)
    nf = np.concatenate([vel, onehot], 1)
    eng.forward(nf, ef); eng.forward(nf, ef); eng.forward(nf, ef)
    eng.set_static(onehot, ef, np.ones(N, np.float32))
    for _ in range(3): eng.ode_step(vel)
    mask = np.nonzero(np.isin(ntype, [0, 5]))[0].astype(np.int32)
    tgt = rng.standard_normal((N, 2)).astype(np.float32)
    for _ in range(4):          # eager, hipGraph capture (forward + backward sequences), two replays; dropped by the next set_graph
        eng.step(nf, ef, tgt, mask)
    eng.rollout("Euler", vel, onehot, ef, 0.0, 0.03, 0.01, 4, dt=0.01)
    eng.rollout("Tsit5", vel, onehot, ef, 0.0, 0.02, 0.01, 3)
    eng.latents_randn(it); eng.processor_steps_dev(15); eng.processor_steps_dev(15); eng.processor_steps_dev(15)
    if it % 20 == 19: log.append(used()); print("iteration %d: device %.1f MB in use, host RSS %.1f MB" % (it + 1, log[-1], _proc.memory_info().rss / 1e6), flush=True)
assert log[-1] - log[1] < 16.0, log
print("no growth after warm-up:", log)
