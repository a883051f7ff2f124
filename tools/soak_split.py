"""Soak of the split path: mid-size meshes of changing size (the ring kernel in four- and eight-wave blocks, the cooperative kernels below
them), parameter changes, processor passes and training steps in one process; device memory must stop growing after the first rounds
and every pass must stay finite.  python tools/soak_split.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np, mgn_amd, bench, psutil

proc = psutil.Process()
def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 1e6

ps = bench.glorot_params()
meshes = [mgn_amd.synth.mesh_1m(7 + i, nx, nx) for i, nx in enumerate((90, 128, 180, 260, 72))]
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
rng = np.random.default_rng(0)
log = []
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for it in range(n_it):
    pos, s, r = meshes[it % len(meshes)]
    N, E = pos.shape[0], s.size
    eng.set_params(ps * (1.0 + 1e-3 * (it % 5)))
    eng.set_graph(s, r, N)
    eng.latents_randn(it)
    for _ in range(3):
        eng.processor_steps_dev(15)
    v, e = eng.latents_export()
    assert np.isfinite(v).all() and np.isfinite(e).all(), it
    if it % 5 == 0:
        nf = rng.standard_normal((N, 9)).astype(np.float32); ef = rng.standard_normal((E, 3)).astype(np.float32)
        tgt = rng.standard_normal((N, 2)).astype(np.float32)
        gs, loss = eng.step(nf, ef, tgt, np.arange(0, N, 3, dtype=np.int32))
        assert np.isfinite(loss) and np.isfinite(gs).all()
    log.append((used(), proc.memory_info().rss / 1e6))
    if it % 10 == 9:
        print(f"it {it + 1}: device {log[-1][0]:.0f} MB, host rss {log[-1][1]:.0f} MB", flush=True)
half = len(log) // 2
print("device MB first/half/last:", round(log[0][0]), round(log[half][0]), round(log[-1][0]))
print("host MB half/last:", round(log[half][1]), round(log[-1][1]))
assert log[-1][0] <= log[half][0] + 16, "device memory still growing"
print("soak OK")
