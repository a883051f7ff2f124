"""Run-to-run determinism of the processor kernels: the same latents through the same 15 steps must come out bit-identical every time (a
race in a kernel's LDS ring -- a window overwritten before its last reader, a parked row read before it is written -- shows up as a
difference between repeats long before it shows up against a tolerance).  Meshes from 8 k to 1 M nodes (the 16-row kernels, the ring
kernels in four- and eight-wave blocks, the fused node kernel).  python tools/soak_determinism.py [repeats]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgn_amd, bench

ps = bench.glorot_params()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for seed, nx in ((3, 90), (4, 128), (5, 180), (6, 354), (7, 1000)):
    pos, s, r = mgn_amd.synth.mesh_1m(seed, nx, nx)
    N, E = pos.shape[0], s.size
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    ref = None
    for k in range(reps):
        eng.latents_randn(11)
        eng.processor_steps_dev(15)
        v, e = eng.latents_export()
        sig = (zlib.crc32(v.tobytes()), zlib.crc32(e.tobytes()))
        if ref is None:
            ref = sig
            assert np.isfinite(v).all() and np.isfinite(e).all()
        elif sig != ref:
            bad += 1
            print(f"N={N}: repeat {k} differs from repeat 0", flush=True)
    fam = eng.debug_last_families() if hasattr(eng, "debug_last_families") else None
    print(f"N={N} E={E}: {reps} repeats, signature {ref[0]:08x}/{ref[1]:08x}" + (f", kernels {fam}" if fam else ""), flush=True)
    del eng
print("determinism OK" if bad == 0 else f"{bad} DIFFERENCES")
sys.exit(1 if bad else 0)
