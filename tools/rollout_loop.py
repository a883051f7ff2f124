"""The cfg-5 shaped rollout (M-cyl, Tsit5, 101 saves) as a profiling target:
rocprofv3 --kernel-trace --stats -d out -- python3 tools/rollout_loop.py [Euler|Tsit5] [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import mgn_amd
import bench

alg = sys.argv[1] if len(sys.argv) > 1 else "Tsit5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pos, cells, ntype, vel = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
N = pos.shape[0]
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params())
eng.set_graph(s, r, N)
rng = np.random.default_rng(0)
onehot = np.eye(7, dtype=np.float32)[np.clip(ntype, 0, 6)]
ef = np.concatenate([pos[s] - pos[r], np.linalg.norm(pos[s] - pos[r], axis=1, keepdims=True)], axis=1).astype(np.float32)
x0 = vel.astype(np.float32) if vel is not None else rng.standard_normal((N, 2)).astype(np.float32)
vm = np.ones(N, np.float32)
for _ in range(reps):
    sol, st = eng.rollout(alg, x0, onehot, ef, 0.0, 1.0, 0.01, 101, dt=0.01 if alg == "Euler" else None, val_mask=vm) if alg == "Euler" else \
        eng.rollout(alg, x0, onehot, ef, 0.0, 1.0, 0.01, 101, val_mask=vm, abstol=1e-6, reltol=1e-3)
print(st)
