"""Locality-restoring node numbering inside mgn_set_graph (csrc/graph_host.cpp): a mesh that arrives with scattered node labels -- DeepMind's
trajectories carry arbitrary ones and create_base_graph passes them through, reference src/graph.jl:30-36 -- is numbered in breadth-first
order inside the engine.  Nothing of it may show at the boundary: every entry point takes and returns the caller's order, and the
results are the oracle's on the caller's graph."""
from importlib import import_module

import numpy as np
import pytest
import torch

import mgn_oracle as orc
import mgn_amd
from mgn_amd import synth
from util import TOL_15, TOL_ROLLOUT, cfg_dict, engine_for, make_params, rel_max, renumbered, scatter_labels, set_renumber

pytestmark = pytest.mark.gpu


def _mesh(nx=40, ny=33, seed=9):
    pos, cells = synth.grid_mesh(nx, ny, seed)
    s, r = synth.cells_to_edges(cells)
    return scatter_labels(pos, s, r, seed=2)


def test_forward_processor_and_training_on_scattered_labels():
    cfg = cfg_dict(mps=3)
    pos, s, r, perm = _mesh()
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    rng = np.random.default_rng(0)
    nf, ef = rng.standard_normal((N, 9)).astype(np.float32), rng.standard_normal((E, 3)).astype(np.float32)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    assert renumbered(eng) and not np.array_equal(eng.owned_nodes(), np.arange(N))
    assert rel_max(eng.forward(nf, ef), orc.forward(ps, cfg, nf, ef, s, r)) <= TOL_15
    v0, e0 = rng.standard_normal((N, 128)).astype(np.float32), rng.standard_normal((E, 128)).astype(np.float32)
    v1, e1 = eng.processor_steps(v0, e0, 3)
    rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, 3)
    assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15
    # step! : target rows and mask entries are the caller's node ids
    target = rng.standard_normal((N, 2)).astype(np.float32)
    mask = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.int32)
    gs, loss = eng.step(nf, ef, target, mask)
    rgs, rloss = orc.step_grads(ps, cfg, nf, ef, s, r, target, mask)
    assert abs(loss - rloss) <= 1e-5 * abs(rloss) and np.linalg.norm(gs - rgs) <= 5e-3 * np.linalg.norm(rgs)
    gs1, loss1 = eng.step(nf, ef, target, mask + 1, mask_index_base=1)       # Julia indices
    assert loss1 == loss and np.array_equal(gs1, gs)
    # the pullback of the model call: the cotangent of nf comes back in the caller's row order
    ybar = rng.standard_normal((N, 2)).astype(np.float32)
    nfbar, gv, out = eng.forward_vjp(nf, ef, ybar, want_out=True)
    rout, rgv, rnf = orc.model_vjp(ps, cfg, nf, ef, s, r, lambda o: ybar.astype(np.float64))
    assert rel_max(out, rout) <= TOL_15
    assert np.linalg.norm(nfbar - rnf) <= 5e-3 * np.linalg.norm(rnf) and np.linalg.norm(gv - rgv) <= 5e-3 * np.linalg.norm(rgv)
    # the same numbers as on the coherently labelled mesh (policy 0 keeps the caller's order): summation order only
    old = set_renumber(0)
    try:
        e0_ = engine_for(cfg)
        e0_.set_params(ps)
        e0_.set_graph(s, r, N)
        assert not renumbered(e0_)
        assert rel_max(e0_.forward(nf, ef), eng.forward(nf, ef)) <= 1e-5
    finally:
        set_renumber(old)


def test_rhs_vjp_and_rollout_on_scattered_labels():
    cfg = cfg_dict(mps=2)
    pos, s, r, perm = _mesh(24, 20, 4)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    rng = np.random.default_rng(3)
    node_type = rng.choice([0, 0, 0, 1, 4, 5, 6], size=N).astype(np.int32)
    onehot = orc.one_hot(node_type, 7, 0).astype(np.float32)
    ef_raw = orc.edge_features(pos, s, r).astype(np.float32)
    x = (rng.standard_normal((N, 2)) * 0.3 + 1.0).astype(np.float32)
    lam = rng.standard_normal((N, 2)).astype(np.float32)
    n_norm = orc.NormMeanStd(np.array([1.0, 0.9]), np.array([0.31, 0.27]))
    t_norm = orc.NormMinMax(0.0, 1.0)
    e_norm = orc.NormMeanStd(ef_raw.mean(0), ef_raw.std(0))
    o_norm = orc.NormMeanStd(np.array([0.01, -0.02]), np.array([0.5, 0.4]))
    vm = np.isin(node_type, [0, 5]).astype(np.float32)
    ns, nsh = n_norm.affine(2)
    ts, tsh = t_norm.affine(7)
    es, esh = e_norm.affine(3)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    assert renumbered(eng)
    eng.set_norms(node=(np.concatenate([ns, ts]), np.concatenate([nsh, tsh])), edge=(es, esh), out=(o_norm.std, o_norm.mean))
    xbar, gs, dxdt = eng.ode_vjp(x, onehot, ef_raw, lam, val_mask=vm, want_dxdt=True)
    rx, rg, rf = orc.ode_vjp(ps, cfg, x, onehot, ef_raw, s, r, n_norm, t_norm, e_norm, o_norm, vm, lam)
    assert rel_max(dxdt, rf) <= TOL_15
    assert np.linalg.norm(xbar - rx) <= 5e-3 * np.linalg.norm(rx) and np.linalg.norm(gs - rg) <= 5e-3 * np.linalg.norm(rg)
    # native rollout with inflow rows (mask and frames in the caller's order)
    inflow = np.repeat((node_type == 1)[:, None], 2, 1)
    gt = (rng.standard_normal((7, N, 2)) * 0.3 + 1.0).astype(np.float32)
    dt = 0.01

    def rhs(xx, t):
        return orc.ode_rhs(ps, cfg, xx, onehot, ef_raw, s, r, n_norm, t_norm, e_norm, o_norm, vm[:, None])

    ref = orc.euler_rollout(rhs, x, dt, 6, inflow, gt)
    sol, st = eng.rollout("Euler", x, onehot, ef_raw, 0.0, 6 * dt, dt, 7, dt=dt, val_mask=vm, inflow_mask=inflow[:, 0], inflow_data=gt,
                          inflow_rule="tolerant")
    assert np.linalg.norm(sol - ref) / np.linalg.norm(ref) <= TOL_ROLLOUT


@pytest.mark.parametrize("P", [2, 3])
def test_partitions_of_a_scattered_mesh(P):
    halo = import_module("mgn_amd.halo")
    cfg = cfg_dict(mps=3)
    pos, s, r, perm = _mesh(60, 50, 5)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    rng = np.random.default_rng(1)
    v0, e0 = rng.standard_normal((N, 128)).astype(np.float32), rng.standard_normal((E, 128)).astype(np.float32)
    stream = torch.cuda.current_stream().cuda_stream
    engs = []
    for k in range(P):
        e = engine_for(cfg, rank=k, nranks=P)
        e.set_stream(stream)
        e.set_params(ps)
        e.set_graph(s, r, N, mesh_pos=pos)
        e.latents_import(v0, e0)
        engs.append(e)
    assert all(renumbered(e) for e in engs) and all(e.n_halo > 0 for e in engs)
    mgn_amd.run_processor_staged(engs, halo.LoopbackExchange(engs, torch.device("cuda")), 3)
    torch.cuda.synchronize()
    v, e_ = np.zeros((N, 128), np.float32), np.zeros((E, 128), np.float32)
    for g in engs:
        g.latents_export(v, e_)
    rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, 3)
    assert rel_max(v, rv) <= TOL_15 and rel_max(e_, re) <= TOL_15


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_two_edge_sets_on_scattered_labels(dtype):
    """A cloth mesh (mesh + world edges) under DeepMind-style arbitrary node labels: the handle is renumbered like a one-set handle (the
    numbering follows the mesh set alone, so installing / re-searching the world set keeps V and the mesh latents valid); host-installed
    world edges arrive in the caller's ids, the device search runs on positions gathered into the engine's order and exports the caller's ids."""
    m = synth.mesh_flag(3, 30, 24, radius=0.06)
    N = m["mesh_pos"].shape[0]
    perm = np.random.default_rng(5).permutation(N).astype(np.int32)            # perm[old] = new label
    inv = np.argsort(perm)
    s, r, s2, r2 = perm[m["s"]], perm[m["r"]], perm[m["s2"]], perm[m["r2"]]
    wpos = m["world_pos"][inv]
    cfg = dict(Fn=12, Fe=7, O=3, L=128, hidden_layers=2, mps=3, Fe2=4)
    ps = orc.init_params(12, 7, 3, 128, 2, 3, 5, 0.1, Fe2=4)
    rng = np.random.default_rng(4)
    nf = rng.standard_normal((N, 12)).astype(np.float32)
    ref = orc.forward(ps, cfg, nf, m["ef"], s, r, set2=(m["ef2"], s2, r2))
    tol = TOL_15 if dtype == "f32" else 3e-2
    err = (lambda a: rel_max(a, ref)) if dtype == "f32" else (lambda a: float(np.linalg.norm(a - ref) / np.linalg.norm(ref)))
    eng = mgn_amd.Engine(12, 7, 3, 128, 2, 3, Fe2=4, dtype=dtype)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    assert renumbered(eng)
    eng.set_edge_set(1, s2, r2)
    assert renumbered(eng) and np.array_equal(np.sort(eng.owned_nodes()), np.arange(N))
    eng.set_edge_features(1, m["ef2"])
    assert err(eng.forward(nf, m["ef"])) <= tol
    # device search on the same handle: the same SET of world edges in the caller's ids (receiver-major in the engine's numbering, so the
    # order differs from the host search's), features computed on the device, the same model output
    assert eng.world_edges_dev(1, wpos, 0.06) == s2.size
    sd, rd = eng.edge_set_export(1)
    key = lambda a, b: np.sort(a.astype(np.int64) * N + b)
    assert np.array_equal(key(sd, rd), key(s2, r2))
    assert err(eng.forward(nf, m["ef"])) <= tol
    if dtype == "f32":    # training on the renumbered two-set handle (host-installed sets keep their lists)
        eng.set_edge_set(1, s2, r2)
        eng.set_edge_features(1, m["ef2"])
        target = rng.standard_normal((N, 3)).astype(np.float32)
        mask = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.int32)
        gs, loss = eng.step(nf, m["ef"], target, mask)
        rgs, rloss = orc.step_grads(ps, cfg, nf, m["ef"], s, r, target, mask, set2=(m["ef2"], s2, r2))
        assert abs(loss - rloss) <= 1e-5 * abs(rloss) and np.linalg.norm(gs - rgs) <= 5e-3 * np.linalg.norm(rgs)
    eng.close()
