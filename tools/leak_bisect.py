"""Host RSS growth per entry point (100 repetitions each): which call of the soak loop (tools/leak_check.py) leaks host memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np, mgn_amd, bench, psutil, gc
proc = psutil.Process()
rss = lambda: proc.memory_info().rss / 1e6
ps = bench.glorot_params()
pos, cells, ntype, vel = mgn_amd.synth.mesh_cyl(1, 1500)
s, r = mgn_amd.synth.cells_to_edges(cells)
rel = pos[s] - pos[r]
ef = np.concatenate([rel, np.linalg.norm(rel, axis=1, keepdims=True)], 1).astype(np.float32)
onehot = np.eye(7, dtype=np.float32)[ntype]
N = pos.shape[0]
nf = np.concatenate([vel, onehot], 1)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(ps); eng.set_graph(s, r, N)
mask = np.nonzero(np.isin(ntype, [0, 5]))[0].astype(np.int32)
tgt = np.random.default_rng(0).standard_normal((N, 2)).astype(np.float32)
def op_params(): eng.set_params(ps)
def op_graph(): eng.set_graph(s, r, N)
def op_graph_fwd(): eng.set_graph(s, r, N); eng.forward(nf, ef); eng.forward(nf, ef); eng.forward(nf, ef)
def op_graph_proc(): eng.set_graph(s, r, N); eng.latents_randn(1); eng.processor_steps_dev(15); eng.processor_steps_dev(15); eng.processor_steps_dev(15)
def op_graph_step():
    eng.set_graph(s, r, N)
    for _ in range(4): eng.step(nf, ef, tgt, mask)
def op_graph_rhs():
    eng.set_graph(s, r, N); eng.set_static(onehot, ef, np.ones(N, np.float32))
    for _ in range(3): eng.ode_step(vel)
def op_graph_rollout(): eng.set_graph(s, r, N); eng.rollout("Euler", vel, onehot, ef, 0.0, 0.03, 0.01, 4, dt=0.01)
def op_fwd_only(): eng.forward(nf, ef)
for name, op in (("set_params", op_params), ("set_graph", op_graph), ("forward only (replay)", op_fwd_only), ("set_graph + 3 forward (capture)", op_graph_fwd),
                 ("set_graph + 3 processor passes (capture)", op_graph_proc), ("set_graph + 4 step (capture)", op_graph_step),
                 ("set_graph + set_static + 3 ode_step (capture)", op_graph_rhs), ("set_graph + rollout", op_graph_rollout)):
    for _ in range(5): op()
    gc.collect(); a = rss()
    for _ in range(100): op()
    gc.collect(); b = rss()
    print("%-50s %+8.2f MB per 100 calls" % (name, b - a), flush=True)
