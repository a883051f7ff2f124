"""us per processor step over cylinder-like meshes of 1.9 k .. 4.3 k nodes (the upper half of the 16-row kernels' range): python tools/small_sweep.py"""
import sys, time; sys.path.insert(0,".")
import torch, numpy as np, mgn_amd, bench
for n in (1900, 2050, 2100, 2300, 2700, 3200, 3900, 4300):
    pos, cells, _, _ = mgn_amd.synth.mesh_cyl(1234, n); s, r = mgn_amd.synth.cells_to_edges(cells)
    eng = mgn_amd.Engine(9,3,2,128,2,15); ps = bench.glorot_params(); eng.set_params(ps); eng.set_graph(s,r,pos.shape[0]); eng.latents_randn(1)
    for _ in range(5): eng.processor_steps_dev(15)
    eng.synchronize(); t=time.perf_counter()
    for _ in range(50): eng.processor_steps_dev(15)
    eng.synchronize(); dt=(time.perf_counter()-t)/750*1e6
    print("N",pos.shape[0],"E",s.size,"us/step %.1f"%dt, flush=True)
