"""us per processor step over mesh sizes between the cylinder mesh and M-1M, for each kernel family (MGN_KERNEL_PATH via
mgn_debug_kernel_path: 0 auto, 1 LDS-resident persistent, 2 all-streaming, 3 cooperative): where are the crossovers?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgn_amd, bench
lib = mgn_amd.load()
lib.mgn_debug_kernel_path.restype = C.c_int; lib.mgn_debug_kernel_path.argtypes = [C.c_int]
ps = bench.glorot_params()
sizes = [int(a) for a in sys.argv[1:]] or [45, 64, 90, 128, 180, 256, 360]
print("%6s %8s %9s | %9s %9s %9s %9s  (us per processor step)" % ("nx", "N", "E", "auto", "resident", "streaming", "coop"))
for nx in sizes:
    pos, s, r = mgn_amd.synth.mesh_1m(1234, nx, nx)
    row = []
    for path in (0, 1, 2, 3):
        lib.mgn_debug_kernel_path(path)
        eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
        eng.set_params(ps); eng.set_graph(s, r, pos.shape[0]); eng.latents_randn(1)
        for _ in range(3): eng.processor_steps_dev(15)
        eng.synchronize(); reps = max(3, int(3e5 / s.size)); t = time.perf_counter()
        for _ in range(reps): eng.processor_steps_dev(15)
        eng.synchronize(); row.append((time.perf_counter() - t) / (15 * reps) * 1e6)
        eng.close()
    lib.mgn_debug_kernel_path(0)
    print("%6d %8d %9d | %9.1f %9.1f %9.1f %9.1f" % (nx, pos.shape[0], s.size, *row), flush=True)
