"""PCIe-inclusive rate of the host-buffer entry point mgn_processor_steps on M-1M (15 steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgn_amd, bench
pos, s, r = mgn_amd.synth.mesh_1m(1234)
N, E = pos.shape[0], s.size
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, N)
rng = np.random.default_rng(0)
v = rng.standard_normal((N, 128), dtype=np.float32); e = rng.standard_normal((E, 128), dtype=np.float32)
eng.processor_steps(v[:], e[:], 1)
t = time.time(); v1, e1 = eng.processor_steps(v, e, 15); dt = time.time() - t
print("mgn_processor_steps(host v,e; 15 steps): %.1f ms  -> %.3g edges/s PCIe-inclusive (%.2f GB each way)" % (dt * 1e3, E * 15 / dt, (v.nbytes + e.nbytes) / 1e9))
