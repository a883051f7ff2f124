"""Diagnostic: per-tile phase durations of the split edge kernels (k_edge_ring, MGN_FP32_SPLIT=1; k_edge_split2, =2) from s_memtime
stamps.  Needs a library built with -DMGN_DIAG_STAMPS: python tools/build_variant.py stamps -DMGN_DIAG_STAMPS, then
MGN_LIB_PATH=<repo>/meshgraphnets.jl_amd/lib/variants/stamps.so python tools/diag_stamps_split.py [nx]."""
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
f = eng.lib.mgn_debug_edge_stamps; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
assert f(eng.h, 1, out.ctypes.data_as(C.c_void_p)) == 0
st = out[:4 * 8 * 24 * 8].reshape(4, 8, 24, 8).astype(np.int64)
names = ["L1", "tab", "L2", "tab", "L3", "reload e+LN", "resid+store", "scan+tails+turnover(to next start)"]
if os.environ.get("MGN_SPLIT_F16", "1") != "0":   # k_edge_ring_h (the default): its stamps
    names = ["rowmax + Q scale + L1 + P", "rowmax 2", "L2", "bound 3", "L3", "reload e + bias + LN", "resid + store e", "scan+tails+turnover(to next start)"]
if os.environ.get("MGN_RING_EPI"):      # library built with -DMGN_RING_EPI_STAMPS as well: the epilogue of k_edge_ring in detail
    names = ["chains", "LN", "resid+store e", "next e req + scan setup", "scan", "tail stores", "Q request", "to next tile top"]
for b in range(2):
    ext = np.concatenate([st[b, :, 2:20, :8], st[b, :, 3:21, 0:1]], axis=-1)
    d = np.diff(ext, axis=-1)   # [wave][tile][phase]
    print(f"block {b}: mean cycles per phase (tiles 2..19), rows = waves:")
    for w in range(8):
        print("  wave", w, {n: int(d[w, :, i].mean()) for i, n in enumerate(names)}, "tile period", int(np.diff(st[b, w, 2:20, 0]).mean()))
