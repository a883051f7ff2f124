#!/usr/bin/env python3
"""Generates the committed golden vectors from the float64 oracle (oracle/mgn_oracle.py).

The reference itself pins nothing (test/runtests.jl:11-19 is Aqua-only) and cannot be run here (no julia,
GraphNetCore.jl not vendored), so these vectors pin the build's own MGN-spec v1: they are produced ONCE by the
float64 NumPy oracle and every implementation (C restatement, HIP engine) is compared against them.
Run from the repo root:  python tests/golden/gen_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mgn_amd  # noqa: E402
import mgn_oracle as orc  # noqa: E402

SEED = 1234


def mesh():
    pos, cells = mgn_amd.synth.grid_mesh(8, 6, SEED)          # N = 48
    s, r = orc.triangles_to_edges(cells)                       # reference order: first occurrence, two-way
    return pos, cells, s.astype(np.int32), r.astype(np.int32)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gold(L, mps, name, snaps):
    pos, cells, s, r = mesh()
    N, E = pos.shape[0], s.size
    cfg = dict(Fn=9, Fe=3, O=2, L=L, hidden_layers=2, mps=mps)
    ps = orc.init_params(9, 3, 2, L, 2, mps, seed=SEED, ln_jitter=0.1)
    rng = np.random.default_rng(SEED)
    nf = rng.standard_normal((N, 9)).astype(np.float32)
    ef = rng.standard_normal((E, 3)).astype(np.float32)
    out, lat = orc.forward(ps, cfg, nf, ef, s, r, return_latents=True)
    d = dict(L=L, mps=mps, seed=SEED, jitter=0.1, params_sha256=sha(ps), senders=s, receivers=r, nf=nf, ef=ef, out=out)
    for k in snaps:
        d[f"v_after_{k}"] = lat[k][0].astype(np.float32)
        d[f"e_after_{k}"] = lat[k][1].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "N", N, "E", E, "out[0]", out[0])


def gold_rollout():
    """GOLD-D: 10-step fixed-dt Euler rollout through the ode_func_eval -> ode_step wrapper
    (reference src/solve.jl:147-158,188-219) on a cylinder_flow-shaped toy problem."""
    pos, cells, s, r = mesh()
    N, E = pos.shape[0], s.size
    L, mps = 128, 3
    cfg = dict(Fn=9, Fe=3, O=2, L=L, hidden_layers=2, mps=mps)
    ps = orc.init_params(9, 3, 2, L, 2, mps, seed=SEED + 1, ln_jitter=0.1)
    rng = np.random.default_rng(SEED + 1)
    node_type = rng.choice([0, 0, 0, 1, 4, 5, 6], size=N).astype(np.int32)
    onehot = orc.one_hot(node_type, 7, 0)
    ef_raw = orc.edge_features(pos, s, r)
    x0 = rng.standard_normal((N, 2)) * 0.3 + 1.0
    gt = rng.standard_normal((11, N, 2)) * 0.3 + 1.0          # "data" used for the inflow overwrite
    n_norm = orc.NormMeanStd(np.array([1.0, 0.9]), np.array([0.31, 0.27]))
    t_norm = orc.NormMinMax(0.0, 1.0)
    e_norm = orc.NormMeanStd(ef_raw.mean(0), ef_raw.std(0))
    o_norm = orc.NormMeanStd(np.array([0.01, -0.02]), np.array([0.5, 0.4]))
    val_mask = np.isin(node_type, [0, 5]).astype(np.float64)[:, None]     # types_updated = [0, 5]
    inflow = np.repeat((node_type == 1)[:, None], 2, 1)                   # literal 1: src/MeshGraphNets.jl:593
    dt = 0.01

    def rhs(x, t):
        k = int(np.floor(t / dt + 1e-9))
        return orc.ode_rhs(ps, cfg, x, onehot, ef_raw, s, r, n_norm, t_norm, e_norm, o_norm, val_mask, inflow, gt[k])

    xs = orc.euler_rollout(rhs, x0, dt, 10, inflow, gt)
    ns, nsh = n_norm.affine(2)
    ts, tsh = t_norm.affine(7)
    es, esh = e_norm.affine(3)
    np.savez_compressed(os.path.join(HERE, "gold_d_rollout.npz"), L=L, mps=mps, seed=SEED + 1, jitter=0.1,
                        params_sha256=sha(ps), senders=s, receivers=r, mesh_pos=pos, node_type=node_type, x0=x0, gt=gt,
                        ef_raw=ef_raw.astype(np.float32), node_scale=np.concatenate([ns, ts]), node_shift=np.concatenate([nsh, tsh]),
                        edge_scale=es, edge_shift=esh, out_scale=o_norm.std, out_shift=o_norm.mean, val_mask=val_mask[:, 0],
                        inflow_mask=inflow, dt=dt, xs=xs, dxdt0=rhs(x0, 0.0))
    print("gold_d_rollout", xs.shape, xs[-1, 0])


def gold_two_sets():
    """GOLD-C: two edge sets, flag_simple-shaped widths (Fn = 3 velocity + 9 one-hot, mesh Fe = 7,
    world Fe2 = 4, O = 3), L = 128, mps = 15, on a 12 x 10 folded cloth patch.  fp32 engines are held to the fp32
    tolerance, the bf16 mode to its band."""
    m = mgn_amd.synth.mesh_flag(SEED + 2, 12, 10, radius=0.13)
    N, E, E2 = m["mesh_pos"].shape[0], m["s"].size, m["s2"].size
    L, mps = 128, 15
    cfg = dict(Fn=12, Fe=7, O=3, L=L, hidden_layers=2, mps=mps, Fe2=4)
    ps = orc.init_params(12, 7, 3, L, 2, mps, seed=SEED + 2, ln_jitter=0.1, Fe2=4)
    rng = np.random.default_rng(SEED + 2)
    onehot = orc.one_hot(m["node_type"], 9, 0)
    nf = np.concatenate([m["velocity"], onehot], 1).astype(np.float32)
    # features as the normalisers would hand them over (unit scale), from the synthetic geometry
    ef = ((m["ef"] - m["ef"].mean(0)) / m["ef"].std(0)).astype(np.float32)
    ef2 = ((m["ef2"] - m["ef2"].mean(0)) / m["ef2"].std(0)).astype(np.float32)
    out, lat = orc.forward(ps, cfg, nf, ef, m["s"], m["r"], return_latents=True, set2=(ef2, m["s2"], m["r2"]))
    d = dict(L=L, mps=mps, seed=SEED + 2, jitter=0.1, params_sha256=sha(ps), senders=m["s"], receivers=m["r"],
             senders2=m["s2"], receivers2=m["r2"], nf=nf, ef=ef, ef2=ef2, out=out)
    for k in (1,):
        d[f"v_after_{k}"] = lat[k][0].astype(np.float32)
        d[f"e_after_{k}"] = lat[k][1].astype(np.float32)
        d[f"e2_after_{k}"] = lat[k][2].astype(np.float32)
    d["v_after_15"] = lat[15][0].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "gold_c_two_sets.npz"), **d)
    print("gold_c_two_sets N", N, "E", E, "E2", E2, "out[0]", out[0], "rng", rng.integers(10))


def cyl_rollout_problem():
    """The cfg-5 problem (BASELINE.json configs[4]: cylinder_flow 100-step rollout) on M-cyl, rebuilt from seeds by the
    generator and by the GPU test alike; nothing but the reference solution is stored."""
    pos, cells, ntype, vel = mgn_amd.synth.mesh_cyl(SEED, 2000)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    N = pos.shape[0]
    cfg = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=2, mps=15)
    ps = orc.init_params(9, 3, 2, 128, 2, 15, seed=SEED + 5, ln_jitter=0.05)
    onehot = orc.one_hot(ntype, 7, 0)
    ef_raw = orc.edge_features(pos, s, r)
    x0 = vel.astype(np.float64)
    dt, nsteps = 0.01, 100
    tt = np.arange(nsteps + 1)[:, None, None] * dt
    gt = x0[None] * (1.0 + 0.05 * np.sin(2 * np.pi * tt))                 # inflow "data" frames, one per save point
    n_norm = orc.NormMeanStd(x0.mean(0), x0.std(0))
    t_norm = orc.NormMinMax(0.0, 1.0)
    e_norm = orc.NormMeanStd(ef_raw.mean(0), ef_raw.std(0))
    o_norm = orc.NormMeanStd(np.array([0.0, 0.0]), np.array([0.5, 0.5]))
    val_mask = np.isin(ntype, [0, 5]).astype(np.float64)[:, None]         # types_updated = [0, 5]
    inflow = np.repeat((ntype == 1)[:, None], 2, 1)                       # literal 1: src/MeshGraphNets.jl:593
    return dict(pos=pos, s=s, r=r, N=N, cfg=cfg, ps=ps, onehot=onehot, ef_raw=ef_raw, x0=x0, dt=dt, nsteps=nsteps, gt=gt,
                n_norm=n_norm, t_norm=t_norm, e_norm=e_norm, o_norm=o_norm, val_mask=val_mask, inflow=inflow, ntype=ntype)


def gold_cyl_rollout(tsit5=True):
    """GOLD-E: the 100-step rollout of BASELINE.json configs[4] at its real size (M-cyl: N = 2000, E = 11 954, L = 128, 15
    steps): fixed-step Euler and adaptive Tsit5 with 101 saves, float64 oracle; every 10th save point is stored."""
    p = cyl_rollout_problem()
    dt, ns = p["dt"], p["nsteps"]

    def rhs(x, t):
        k = min(int(np.floor(t / dt + 1e-6)), ns)
        return orc.ode_rhs(p["ps"], p["cfg"], x, p["onehot"], p["ef_raw"], p["s"], p["r"], p["n_norm"], p["t_norm"], p["e_norm"],
                           p["o_norm"], p["val_mask"], p["inflow"], p["gt"][k])

    xs = orc.euler_rollout(rhs, p["x0"], dt, ns, p["inflow"], p["gt"])
    d = dict(seed=SEED, params_sha256=sha(p["ps"]), senders_sha256=sha(p["s"]), receivers_sha256=sha(p["r"]), N=p["N"], E=p["s"].size,
             dt=dt, nsteps=ns, every=10, euler=xs[::10].astype(np.float32), euler_first=xs[1].astype(np.float32))
    print("gold_e euler", xs.shape, xs[-1, 0], flush=True)
    if tsit5:
        def f(x, t):
            k = min(int(np.floor(t / dt + 1e-6)), ns)
            x[p["inflow"]] = p["gt"][k][p["inflow"]]          # in place, like the reference
            return orc.ode_rhs(p["ps"], p["cfg"], x, p["onehot"], p["ef_raw"], p["s"], p["r"], p["n_norm"], p["t_norm"], p["e_norm"],
                               p["o_norm"], p["val_mask"])
        sol, st = orc.tsit5_rollout(f, p["x0"], 0.0, ns * dt, np.arange(ns + 1) * dt, abstol=1e-6, reltol=1e-3)
        d.update(tsit5=sol[::10].astype(np.float32), tsit5_accept=st["n_accept"], tsit5_reject=st["n_reject"], tsit5_rhs=st["n_rhs"])
        print("gold_e tsit5", st, sol[-1, 0], flush=True)
    np.savez_compressed(os.path.join(HERE, "gold_e_cyl_rollout.npz"), **d)


def gold_hidden_layers():
    """GOLD-F: `hidden_layers` other than the example's 2 (reference Args.hidden_layers, src/MeshGraphNets.jl:35-38): h = 1 and h = 3
    on the GOLD-A mesh, L = 32, two processor steps; outputs and the node latents after the last step."""
    pos, cells, s, r = mesh()
    N, E = pos.shape[0], s.size
    rng = np.random.default_rng(SEED + 7)
    nf = rng.standard_normal((N, 9)).astype(np.float32)
    ef = rng.standard_normal((E, 3)).astype(np.float32)
    d = dict(L=32, mps=2, seed=SEED + 7, jitter=0.1, senders=s, receivers=r, nf=nf, ef=ef)
    for hl in (1, 3):
        cfg = dict(Fn=9, Fe=3, O=2, L=32, hidden_layers=hl, mps=2)
        ps = orc.init_params(9, 3, 2, 32, hl, 2, seed=SEED + 7 + hl, ln_jitter=0.1)
        out, lat = orc.forward(ps, cfg, nf, ef, s, r, return_latents=True)
        d[f"h{hl}_params_sha256"] = sha(ps)
        d[f"h{hl}_param_count"] = ps.size
        d[f"h{hl}_out"] = out
        d[f"h{hl}_v_after_2"] = lat[2][0].astype(np.float32)
        print("gold_f h", hl, "params", ps.size, "out[0]", out[0])
    np.savez_compressed(os.path.join(HERE, "gold_f_hidden_layers.npz"), **d)


def gold_ln_variants():
    """GOLD-G: the [GNC-unverified] LayerNorm choices as data (spec_variant; julia/spec_probe.jl tells which
    one the installed GraphNetCore / Lux compute).  Same inputs and parameters as GOLD-A-sized L = 128, mps = 3; three outputs:
    MGN-spec v1, the (std + eps) denominator (engine: ln_mode = 1), and whole-array statistics (oracle only)."""
    pos, cells, s, r = mesh()
    N, E = pos.shape[0], s.size
    L, mps = 128, 3
    cfg = dict(Fn=9, Fe=3, O=2, L=L, hidden_layers=2, mps=mps)
    ps = orc.init_params(9, 3, 2, L, 2, mps, seed=SEED + 7, ln_jitter=0.1)
    rng = np.random.default_rng(SEED + 7)
    nf = rng.standard_normal((N, 9)).astype(np.float32)
    ef = rng.standard_normal((E, 3)).astype(np.float32)
    d = dict(L=L, mps=mps, seed=SEED + 7, jitter=0.1, params_sha256=sha(ps), senders=s, receivers=r, nf=nf, ef=ef)
    for name, mode, dims in (("out_v1", 0, "row"), ("out_std_eps", 1, "row"), ("out_whole_array", 0, "all")):
        orc.LN_MODE, orc.LN_DIMS = mode, dims
        try:
            d[name] = orc.forward(ps, cfg, nf, ef, s, r)
        finally:
            orc.LN_MODE, orc.LN_DIMS = 0, "row"
    np.savez_compressed(os.path.join(HERE, "gold_g_ln_variants.npz"), **d)
    print("gold_g_ln_variants.npz", "v1 vs std_eps", np.abs(d["out_v1"] - d["out_std_eps"]).max(), "v1 vs whole-array",
          np.abs(d["out_v1"] - d["out_whole_array"]).max())


def kats():
    """Known-answer tests stored as data so every implementation reads the same file."""
    tri1 = np.array([[0, 1, 2]], np.int32)
    tri2 = np.array([[0, 1, 2], [1, 3, 2]], np.int32)
    s1, r1 = orc.triangles_to_edges(tri1)
    s2, r2 = orc.triangles_to_edges(tri2)
    pos345 = np.array([[0.0, 0.0], [3.0, 0.0], [3.0, 4.0]])
    ef = orc.edge_features(pos345, np.array([1, 2, 2]), np.array([0, 1, 0]))
    np.savez_compressed(os.path.join(HERE, "kats.npz"), tri1_s=s1, tri1_r=r1, tri2_s=s2, tri2_r=r2,
                        onehot_types=np.array([0, 4, 5, 6]), onehot=orc.one_hot([0, 4, 5, 6], 7, 0), ef345=ef)
    print("kats: tri1 directed", s1.size, "tri2 directed", s2.size, "norms", ef[:, 2])


if __name__ == "__main__":
    if "--hidden-layers" in sys.argv:
        gold_hidden_layers()
        sys.exit(0)
    if "--ln-variants" in sys.argv:
        gold_ln_variants()
        sys.exit(0)
    if "--cyl-rollout" in sys.argv:       # ~40 minutes of float64 NumPy on 8 cores: generated separately
        gold_cyl_rollout()
        sys.exit(0)
    gold(32, 1, "gold_a_L32_mps1.npz", [0, 1])
    gold(128, 15, "gold_b_L128_mps15.npz", [1, 8, 15])
    gold_rollout()
    gold_two_sets()
    gold_hidden_layers()
    gold_ln_variants()
    kats()
