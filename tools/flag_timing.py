"""us per processor step on the M-flag cloth (two edge sets), fp32 and bf16, hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgn_amd, bench
mf = mgn_amd.synth.mesh_flag()
N = mf["mesh_pos"].shape[0]
ps = bench.glorot_params(1234, 12, 7, 3, Fe2=4)
for dt in ("f32", "bf16"):
    eng = mgn_amd.Engine(12, 7, 3, 128, 2, 15, dtype=dt, Fe2=4)
    eng.set_params(ps); eng.set_graph(mf["s"], mf["r"], N); eng.set_edge_set(1, mf["s2"], mf["r2"]); eng.latents_randn(1)
    for _ in range(5): eng.processor_steps_dev(15)
    eng.synchronize(); t = time.perf_counter()
    for _ in range(100): eng.processor_steps_dev(15)
    eng.synchronize(); print(dt, "us/step %.1f" % ((time.perf_counter() - t) / 1500 * 1e6), eng.latents_checksum()["sumsq_v"])
