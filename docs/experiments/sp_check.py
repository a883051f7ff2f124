import sys, ctypes as C, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import torch, mgn_amd, bench
lib = mgn_amd.load()
lib.mgn_debug_edge_sp.restype = C.c_int; lib.mgn_debug_edge_sp.argtypes=[C.c_int]
ps = bench.glorot_params()
for nx in (170, 1000):
    pos, s, r = mgn_amd.synth.mesh_1m(1234, nx, nx)
    N = pos.shape[0]
    res = {}
    for sp in (0, 1):
        lib.mgn_debug_edge_sp(sp)
        eng = mgn_amd.Engine(9,3,2,128,2,15)
        eng.set_params(ps); eng.set_graph(s, r, N); eng.latents_randn(7)
        eng.processor_steps_dev(3); eng.synchronize()
        chk = eng.latents_checksum()
        if nx == 170:
            v, e = eng.latents_export()
            res[sp] = (chk, v, e)
        else:
            res[sp] = (chk,)
        # timing
        for _ in range(2): eng.processor_steps_dev(15)
        eng.synchronize(); t0=time.perf_counter()
        for _ in range(5): eng.processor_steps_dev(15)
        eng.synchronize(); dt=(time.perf_counter()-t0)/75
        eng.profile_enable(True); eng.processor_steps_dev(15); eng.synchronize(); pr=eng.profile_read(); eng.profile_enable(False)
        print(f"nx={nx} sp={sp}: {dt*1e3:.4f} ms/step, edge {pr['edge_step']['avg_ms']:.4f} ms, node {pr['node_step']['avg_ms']:.4f}", chk)
        eng.close()
    print("  checksums equal:", res[0][0]==res[1][0], " latents bitwise equal:" , (np.array_equal(res[0][1],res[1][1]) and np.array_equal(res[0][2],res[1][2])) if nx==170 else "n/a")
