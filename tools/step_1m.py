"""mgn_step on the full M-1M mesh (N = 1 000 000, E = 5 992 002, L = 128, 15 steps): time per training step (activations stored for as many steps as memory holds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch   # before the engine's first HIP call
import numpy as np, mgn_amd, bench
pos, s, r = mgn_amd.synth.mesh_1m(1234)
N, E = pos.shape[0], s.size
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, N)
rng = np.random.default_rng(0)
nf = rng.standard_normal((N, 9), dtype=np.float32); ef = rng.standard_normal((E, 3), dtype=np.float32)
tgt = rng.standard_normal((N, 2), dtype=np.float32); mask = np.arange(0, N, 2, dtype=np.int32)
t = time.time(); gs, loss = eng.step(nf, ef, tgt, mask); t1 = time.time() - t
t = time.time(); gs2, loss2 = eng.step(nf, ef, tgt, mask); t2 = time.time() - t
import ctypes as C
eng.lib.mgn_debug_train_keep_steps.argtypes = [C.c_void_p]
print("processor steps with stored activations:", eng.lib.mgn_debug_train_keep_steps(eng.h), "of 15")
ts = []
for _ in range(3):
    t = time.time(); eng.step(nf, ef, tgt, mask); ts.append(time.time() - t)
print("three more calls: " + " ".join("%.3f" % x for x in ts))
print("N", N, "E", E, "first call %.2f s, second %.3f s; loss %.6f; grads finite %s; deterministic %s; device mem in use %.1f GB"
      % (t1, t2, loss, bool(np.isfinite(gs).all()), bool(np.array_equal(gs, gs2) and loss == loss2),
         (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9))
