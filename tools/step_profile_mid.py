"""us per processor step and per-family kernel times (mgn_profile_*) on an nx x nx slice of the M-1M generator: python tools/step_profile_mid.py NX"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, mgn_amd, bench
nx = int(sys.argv[1])
pos, s, r = mgn_amd.synth.mesh_1m(1234, nx, nx)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, pos.shape[0]); eng.latents_randn(1)
for _ in range(3): eng.processor_steps_dev(15)
eng.synchronize(); t = time.perf_counter()
for _ in range(10): eng.processor_steps_dev(15)
eng.synchronize(); dt = (time.perf_counter() - t) / 150
print("N", pos.shape[0], "E", s.size, "us/step %.1f  ns/edge %.3f" % (dt * 1e6, dt * 1e9 / s.size))
eng.profile_enable(True)
for _ in range(5): eng.processor_steps_dev(15)
eng.synchronize()
print(eng.profile_read())
