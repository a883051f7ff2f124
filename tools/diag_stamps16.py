"""Diagnostic: phase timeline of the 16-row cooperative kernels on the cylinder-sized mesh from s_memtime stamps (variant built
with -DMGN_DIAG_STAMPS, MGN_LIB_PATH pointing at it).  Prints, per kernel, the distribution over blocks of each phase's length and
of the block start / end offsets from the first stamp of the launch (us: s_memrealtime, the 100 MHz counter all CUs share)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import mgn_amd, bench
pos, cells, _, _ = mgn_amd.synth.mesh_cyl(1234, 2000)
s, r = mgn_amd.synth.cells_to_edges(cells)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
eng.set_params(bench.glorot_params()); eng.set_graph(s, r, pos.shape[0]); eng.latents_randn(1)
for _ in range(5): eng.processor_steps_dev(15)
eng.synchronize()
TICK = float(os.environ.get("MGN_TICK_US", "0.01"))
for which, names in (("edge", ["idx, issue loads", "chain1(+gather wait)", "relu+xch1", "chain2", "xch2+chain3", "xch3", "LN+store+scan"]),
                     ("node", ["agg+issue", "chain1v", "chain1a", "xch+chain2", "xch+chain3+xch+LN", "store+xch", "P,Q chains"])):
    f = getattr(eng.lib, f"mgn_debug_{which}_stamps"); f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    for rep in range(2):
        out = np.zeros(32768, np.uint64)
        assert f(eng.h, 1, out.ctypes.data_as(C.c_void_p)) == 0
    st = out.reshape(1024, 4, 8).astype(np.int64)
    live = st[:, 0, 0] > 0
    nb = int(live.sum())
    st = st[:nb]
    print(f"== {which}: {nb} stamped blocks")
    q = lambda v: "min %.2f med %.2f max %.2f" % (v.min() * TICK, np.median(v) * TICK, v.max() * TICK)
    for x in range(8):
        sx = st[x::8]
        t0 = sx[:, :, 0].min()
        start = np.sort(sx[:, :, 0].min(axis=1) - t0) * TICK
        end = np.sort(sx[:, :, 7].max(axis=1) - t0) * TICK
        print(f"  XCD {x}: {sx.shape[0]} blocks; start offsets (us) deciles", np.round(np.quantile(start, np.linspace(0, 1, 11)), 2).tolist(),
              "end: first %.2f median %.2f last %.2f" % (end[0], np.median(end), end[-1]))
    t0 = st[0, :, 0].min()
    d = np.diff(st, axis=-1)
    for i, nme in enumerate(names):
        print("  %-24s %s" % (nme, q(d[:, :, i])))
    print("  whole tile              ", q(st[:, :, 7] - st[:, :, 0]))
    for b in (0, 1, 100):
        if b < st.shape[0]:
            print(f"  block {b} wave timelines (us from launch start):", np.round((st[b] - t0) * TICK, 2).tolist())
    if which == "edge":
        rel = (st - st[:, :, 0:1].min(axis=1, keepdims=True)) * TICK          # [block][wave][slot] from the block's own start
        s1 = rel[:, :, 2].max(axis=1)
        print("  stamp-2 offset (chain 1 may start) by block index, blocks 0..747 step 17:")
        print("   ", [(int(b), round(float(s1[b]), 2)) for b in range(0, nb, 17)])
        for lo in range(0, nb, 128):
            hi = min(nb, lo + 128)
            print("   blocks %3d..%3d: median offsets of stamps 1..7 from the block's start:" % (lo, hi - 1),
                  [round(float(np.median(rel[lo:hi, :, k].max(axis=1))), 2) for k in range(1, 8)])
