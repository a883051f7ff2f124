"""Soak of the small-mesh path (16-row cooperative kernels on the split path): meshes of changing size (one to six row tiles per block),
one and two edge sets, fp32 and bf16 storage; every configuration is run several times from the same latents and must give the same bits
every time, finite.  python tools/soak_small.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np, mgn_amd, bench

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ps = bench.glorot_params()
psf = bench.glorot_params(1234, 12, 7, 3, Fe2=4)
bad = 0
for rnd in range(rounds):
    for n_pts in (400, 900, 1300, 2000, 2300, 3200, 3900):
        pos, cells, _, _ = mgn_amd.synth.mesh_cyl(100 + rnd, n_pts)
        s, r = mgn_amd.synth.cells_to_edges(cells)
        for dt in ("f32", "bf16"):
            eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15, dtype=dt)
            eng.set_params(ps); eng.set_graph(s, r, pos.shape[0])
            ref = None
            for rep in range(4):
                eng.latents_randn(rnd)
                eng.processor_steps_dev(15)
                v, e = eng.latents_export()
                ok = np.isfinite(v).all() and np.isfinite(e).all()
                if ref is None: ref = (v, e)
                same = np.array_equal(v, ref[0]) and np.array_equal(e, ref[1])
                if not (ok and same):
                    bad += 1; print("MISMATCH", rnd, n_pts, dt, rep, ok, same, flush=True)
            eng.close()
    mf = mgn_amd.synth.mesh_flag(seed=rnd, nx=24 + 4 * (rnd % 5), ny=30)
    eng = mgn_amd.Engine(12, 7, 3, 128, 2, 15, Fe2=4)
    eng.set_params(psf); eng.set_graph(mf["s"], mf["r"], mf["mesh_pos"].shape[0]); eng.set_edge_set(1, mf["s2"], mf["r2"])
    ref = None
    for rep in range(4):
        eng.latents_randn(rnd); eng.processor_steps_dev(15)
        v, e = eng.latents_export(); e2 = eng.edge_latents_export(1)
        ok = np.isfinite(v).all() and np.isfinite(e).all() and np.isfinite(e2).all()
        if ref is None: ref = (v, e, e2)
        same = all(np.array_equal(a, b) for a, b in zip((v, e, e2), ref))
        if not (ok and same):
            bad += 1; print("MISMATCH flag", rnd, rep, ok, same, flush=True)
    eng.close()
    if rnd % 5 == 4: print("round", rnd + 1, "ok" if not bad else f"{bad} mismatches", flush=True)
print("soak_small:", "PASS" if not bad else f"FAIL ({bad})")
