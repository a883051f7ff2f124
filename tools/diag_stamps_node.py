"""Diagnostic: per-tile phase durations of k_node_split_h from s_memtime stamps (library built with -DMGN_DIAG_STAMPS:
python tools/build_variant.py stamps split:-DMGN_DIAG_STAMPS; MGN_LIB_PATH=.../variants/stamps.so python tools/diag_stamps_node.py [nx])."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgn_amd, bench
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pos, s, r = mgn_amd.synth.mesh_1m(1234, nx, nx)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, pos.shape[0]); eng.latents_randn(1)
eng.processor_steps_dev(2)
out = np.zeros(32768, np.uint64)
f = eng.lib.mgn_debug_node_stamps; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
assert f(eng.h, 1, out.ctypes.data_as(C.c_void_p)) == 0
st = out[:4 * 8 * 24 * 8].reshape(4, 8, 24, 8).astype(np.int64)
names = ["rowmax + L1 node part", "aggregate load + finish", "L1 aggregate part", "L2", "L3", "V again + LN + residual", "V store", "next V request (to next start)"]
for b in range(2):
    ext = np.concatenate([st[b, :, 2:12, :8], st[b, :, 3:13, 0:1]], axis=-1)
    d = np.diff(ext, axis=-1)
    print(f"block {b}: mean cycles per phase (tiles 2..11), rows = waves:")
    for w in range(8):
        print("  wave", w, {n: int(d[w, :, i].mean()) for i, n in enumerate(names)}, "tile period", int(np.diff(st[b, w, 2:12, 0]).mean()))
