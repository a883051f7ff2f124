"""Diagnostic: phase durations of the cooperative edge kernel (small meshes) from s_memtime stamps (build the library with
-DMGN_DIAG_STAMPS).  Prints microseconds between stamps for the 4 waves of blocks 0..3 (100 MHz counter)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgn_amd, bench
pos, cells, _, _ = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, pos.shape[0]); eng.latents_randn(1)
eng.processor_steps_dev(2)
out = np.zeros(4 * 8 * 24 * 8, np.uint64)
f = eng.lib.mgn_debug_edge_stamps; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
for rep in range(2):
    assert f(eng.h, 1, out.ctypes.data_as(C.c_void_p)) == 0
st = out.reshape(4, 8, 24, 8).astype(np.int64)[:, :4, 0, :]       # [block][wave][slot]
names = ["loads issued", "L1 (wait+chain)", "xch1", "L2+xch2", "L3+xch3", "LN+store", "scan+tails"]
t0 = st[:, :, 0].min()
print("values in units of 100 s_memtime ticks (core clock, ~2.4 GHz: 100 ticks = 0.042 us)")
for b in range(4):
    for w in range(4):
        d = np.diff(st[b, w]) / 100.0
        print(f"block {b} wave {w}: start +{(st[b, w, 0] - t0) / 100.0:5.2f} |", "  ".join(f"{n} {x:5.2f}" for n, x in zip(names, d)), f"| total {(st[b, w, 7] - st[b, w, 0]) / 100.0:5.2f}")
