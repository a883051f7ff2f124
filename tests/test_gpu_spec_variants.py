"""spec_variant switches: what GraphNetCore 0.3 / Lux 0.5 really compute cannot be read here (sources not
vendored, Project.toml:11,15,36,40), so the [GNC-unverified] LayerNorm choice is a create-time flag -- mgn_config.ln_mode -- that
julia/spec_probe.jl tells a maintainer how to set.  The engine with ln_mode = MGN_LN_STD_EPS must match the oracle's LN_MODE = 1 in
every kernel family, reproduce the GOLD-G fixture, and must NOT match MGN-spec v1 (the flag really reaches the kernels)."""
import os

import numpy as np
import pytest
import torch   # noqa: F401

import mgn_oracle as orc
from mgn_amd import synth
from mgn_amd.engine import MgnError
from util import TOL_15, cfg_dict, engine_for, make_params, random_inputs, rel_max, set_fp32_split, set_kernel_path, set_split_f16, small_mesh

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture
def std_eps_oracle():
    orc.LN_MODE = 1
    yield
    orc.LN_MODE = 0


@pytest.mark.parametrize("path", [0, 1, 2, 3, 5])
def test_gold_g_std_eps_in_every_kernel_family(path):
    g = np.load(os.path.join(GOLD, "gold_g_ln_variants.npz"))
    cfg = cfg_dict(L=int(g["L"]), mps=int(g["mps"]))
    ps = orc.init_params(9, 3, 2, cfg["L"], 2, cfg["mps"], seed=int(g["seed"]), ln_jitter=float(g["jitter"]))
    old = set_kernel_path(path)
    try:
        for mode, name in ((1, "out_std_eps"), (0, "out_v1")):
            eng = engine_for(cfg, ln_mode=mode)
            eng.set_params(ps)
            eng.set_graph(g["senders"], g["receivers"], g["nf"].shape[0])
            out = eng.forward(g["nf"], g["ef"])
            assert rel_max(out, g[name]) <= TOL_15, (path, name)
            if mode == 1:
                assert rel_max(out, g["out_v1"]) > 5 * rel_max(out, g[name])      # it is the other LayerNorm
            eng.close()
    finally:
        set_kernel_path(old)


@pytest.mark.parametrize("split", [0, 1, 101], ids=["fp32_mfma", "split_bf16x3", "split_f16x2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_std_eps_on_a_mesh_large_enough_for_the_persistent_kernels(std_eps_oracle, split, dtype):
    """22 500 nodes: the persistent fp32-MFMA kernels (split 0), the split path on three bf16 pieces (k_edge_ring + k_node_split) and on two fp16 pieces
    (101: k_edge_ring_h + k_node_split_h, the default), and the bf16 kernels (their own packed LayerNorm)"""
    if dtype == "bf16" and split:
        pytest.skip("the split path is an fp32 path")
    cfg = cfg_dict(mps=3)
    pos, s, r = synth.mesh_1m(1234, 150, 150)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg, jitter=0.05)
    rng = np.random.default_rng(9)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 3)
    old = set_fp32_split(split % 100)
    oldh = set_split_f16(1 if split == 101 else 0)
    try:
        eng = engine_for(cfg, ln_mode=1, dtype=dtype)
        eng.set_params(ps)
        eng.set_graph(s, r, N)
        v1, e1 = eng.processor_steps(v, e, 3)
    finally:
        set_split_f16(oldh)
        set_fp32_split(old)
    if dtype == "bf16":
        rl2 = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
        assert rl2(v1, rv) <= 3e-2 and rl2(e1, re) <= 3e-2
    else:
        assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15, (rel_max(v1, rv), rel_max(e1, re))


def test_small_sizes_and_latent_widths(std_eps_oracle):
    for L in (32, 64, 128):
        cfg = cfg_dict(L=L, mps=2)
        pos, s, r = small_mesh(11, 7)
        N, E = pos.shape[0], s.size
        ps = make_params(cfg)
        nf, ef = random_inputs(N, E, cfg, 3)
        eng = engine_for(cfg, ln_mode=1)
        eng.set_params(ps)
        eng.set_graph(s, r, N)
        assert rel_max(eng.forward(nf, ef), orc.forward(ps, cfg, nf, ef, s, r)) <= TOL_15, L


def test_training_under_the_std_eps_variant(std_eps_oracle):
    """mgn_step, mgn_forward_vjp and mgn_ode_vjp under ln_mode = MGN_LN_STD_EPS: the pullback of y = d / (sqrt(var) + eps) has one more
    factor than the default's (kappa = (sqrt(var) + eps) / sqrt(var) on the xhat term); against the oracle's reverse mode, which is
    checked against finite differences in both modes on the CPU (tests/test_oracle_golden.py)."""
    for L, nx in ((128, 8), (32, 8), (128, 40)):                    # cooperative kernels, L = 32, and the streaming kernels
        cfg = cfg_dict(L=L, mps=2)
        pos, s, r = small_mesh(nx, nx - 2)
        N, E = pos.shape[0], s.size
        ps = make_params(cfg, jitter=0.1)
        nf, ef = random_inputs(N, E, cfg, 5)
        rng = np.random.default_rng(L + nx)
        target = rng.standard_normal((N, 2)).astype(np.float32)
        mask = rng.choice(N, N // 2, replace=False).astype(np.int32)
        eng = engine_for(cfg, ln_mode=1)
        eng.set_params(ps)
        eng.set_graph(s, r, N)
        gs, loss = eng.step(nf, ef, target, mask)
        g_ref, loss_ref = orc.step_grads(ps, cfg, nf, ef, s, r, target, mask)
        assert abs(loss - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref)), (L, nx)
        assert np.linalg.norm(gs - g_ref) <= 2e-4 * np.linalg.norm(g_ref), (L, nx, np.linalg.norm(gs - g_ref) / np.linalg.norm(g_ref))
        # and it is the other function: the default mode's gradient differs by more than that
        orc.LN_MODE = 0
        g_v1, _ = orc.step_grads(ps, cfg, nf, ef, s, r, target, mask)
        orc.LN_MODE = 1
        assert np.linalg.norm(g_v1 - g_ref) > 10 * np.linalg.norm(gs - g_ref)
        ybar = rng.standard_normal((N, 2)).astype(np.float32)
        nfbar, gps, _ = eng.forward_vjp(nf, ef, ybar)
        _, gp_ref, nfbar_ref = orc.model_vjp(ps, cfg, nf, ef, s, r, lambda out: ybar.astype(np.float64))
        assert np.linalg.norm(gps - gp_ref) <= 2e-4 * np.linalg.norm(gp_ref)
        assert np.linalg.norm(nfbar - nfbar_ref) <= 2e-4 * np.linalg.norm(nfbar_ref)
        eng.close()


def test_bad_modes_are_rejected():
    cfg = cfg_dict(mps=2)
    with pytest.raises(MgnError):
        engine_for(cfg, ln_mode=7)


@pytest.fixture
def whole_array_oracle():
    orc.LN_DIMS = "all"
    yield
    orc.LN_DIMS = "row"


def test_whole_array_layernorm_is_an_engine_mode(whole_array_oracle):
    """mgn_config.ln_dims = MGN_LN_ALL: LayerNorm statistics over the whole (L x rows) output of every MLP -- what Lux 0.5's
    LayerNorm(shape) computes when it is left at dims = Colon() (reference Project.toml:15,40; julia/spec_probe.jl reports it).  The
    engine reproduces the GOLD-G `out_whole_array` fixture, follows the oracle's LN_DIMS = "all" on a 22 500-node mesh (forward of the
    whole model and processor steps, hidden_layers 2 and 3, both ln_mode denominators), in the right-hand side (one-shot and resident forms)
    and in the native rollout driver, is really the other network, serves the device-resident processor entry point through the same
    driver, and refuses the staged per-step calls that belong to the fused kernels."""
    g = np.load(os.path.join(GOLD, "gold_g_ln_variants.npz"))
    cfg = cfg_dict(L=int(g["L"]), mps=int(g["mps"]))
    ps = orc.init_params(9, 3, 2, cfg["L"], 2, cfg["mps"], seed=int(g["seed"]), ln_jitter=float(g["jitter"]))
    eng = engine_for(cfg, ln_dims="all")
    eng.set_params(ps)
    eng.set_graph(g["senders"], g["receivers"], g["nf"].shape[0])
    out = eng.forward(g["nf"], g["ef"])
    assert rel_max(out, g["out_whole_array"]) <= TOL_15
    assert rel_max(out, g["out_v1"]) > 20 * rel_max(out, g["out_whole_array"])
    # the right-hand side the Julia-driven solver calls once per evaluation (one-shot form): build_graph's normalisers, the model,
    # inverse_data, val_mask -- against the oracle's ode_rhs under LN_DIMS = "all"
    rng0 = np.random.default_rng(8)
    N0 = g["nf"].shape[0]
    x0 = (rng0.standard_normal((N0, 2)) * 0.3 + 1.0).astype(np.float32)
    onehot0 = np.eye(7, dtype=np.float32)[rng0.integers(0, 7, N0)]
    vm0 = (rng0.random(N0) < 0.7).astype(np.float32)
    n_norm, t_norm = orc.NormMeanStd(np.array([1.0, 0.9]), np.array([0.31, 0.27])), orc.NormMinMax(0.0, 1.0)
    e_norm, o_norm = orc.NormMeanStd(g["ef"].mean(0), g["ef"].std(0)), orc.NormMeanStd(np.array([0.01, -0.02]), np.array([0.5, 0.4]))
    (ns, nsh), (ts, tsh), (es, esh) = n_norm.affine(2), t_norm.affine(7), e_norm.affine(3)
    eng.set_norms(node=(np.concatenate([ns, ts]), np.concatenate([nsh, tsh])), edge=(es, esh), out=(o_norm.std, o_norm.mean))
    d_ref = orc.ode_rhs(ps, cfg, x0, onehot0, g["ef"], g["senders"], g["receivers"], n_norm, t_norm, e_norm, o_norm, vm0[:, None])
    assert rel_max(eng.ode_step(x0, onehot0, g["ef"], vm0), d_ref) <= TOL_15
    # the resident form (mgn_set_static once per trajectory, then only the state moves): the edge encoder runs on the first evaluation
    eng.set_static(onehot0, g["ef"], vm0)
    assert rel_max(eng.ode_step(x0), d_ref) <= TOL_15
    x1 = (x0 + 0.1 * rng0.standard_normal(x0.shape)).astype(np.float32)
    d_ref1 = orc.ode_rhs(ps, cfg, x1, onehot0, g["ef"], g["senders"], g["receivers"], n_norm, t_norm, e_norm, o_norm, vm0[:, None])
    assert rel_max(eng.ode_step(x1), d_ref1) <= TOL_15
    # the native rollout driver on the same right-hand side (Euler with inflow rows; Tsit5 takes the same path per evaluation)
    inflow = np.repeat((onehot0[:, 1] == 1)[:, None], 2, 1)
    gt = (rng0.standard_normal((5, N0, 2)) * 0.3 + 1.0).astype(np.float32)
    dt = 0.01

    def rhs(xx, t):
        return orc.ode_rhs(ps, cfg, xx, onehot0, g["ef"], g["senders"], g["receivers"], n_norm, t_norm, e_norm, o_norm, vm0[:, None])

    ref = orc.euler_rollout(rhs, x0, dt, 4, inflow, gt)
    sol, _ = eng.rollout("Euler", x0, onehot0, g["ef"], 0.0, 4 * dt, dt, 5, dt=dt, val_mask=vm0, inflow_mask=inflow[:, 0], inflow_data=gt,
                         inflow_rule="tolerant")
    assert np.linalg.norm(sol - ref) / np.linalg.norm(ref) <= 1e-4
    sol5, st5 = eng.rollout("Tsit5", x0, onehot0, g["ef"], 0.0, 2 * dt, dt, 3, val_mask=vm0)
    ref5 = orc.tsit5_rollout(lambda xx, t: rhs(xx, t), x0.astype(np.float64), 0.0, 2 * dt, [0.0, dt, 2 * dt])
    assert np.linalg.norm(sol5 - np.stack(ref5[0])) / np.linalg.norm(np.stack(ref5[0])) <= 1e-3 and st5["n_rhs"] >= 7
    # the device-resident processor entry point (round 6): the resident latents go through the unfused driver as rows and come back --
    # the same bits as mgn_processor_steps on host arrays; the staged per-step calls of the fused kernels still answer MGN_E_UNSUPPORTED
    rngl = np.random.default_rng(12)
    v0 = rngl.standard_normal((N0, 128)).astype(np.float32)
    e0 = rngl.standard_normal((g["ef"].shape[0], 128)).astype(np.float32)
    v_host, e_host = eng.processor_steps(v0.copy(), e0.copy(), 2)
    rv, re_ = orc.processor_steps(ps, cfg, v0, e0, g["senders"], g["receivers"], 2)
    assert rel_max(v_host, rv) <= TOL_15 and rel_max(e_host, re_) <= TOL_15
    eng.latents_import(v0, e0)
    eng.processor_steps_dev(2)
    v_dev, e_dev = eng.latents_export()
    assert np.array_equal(v_dev, v_host) and np.array_equal(e_dev, e_host)
    with pytest.raises(MgnError) as ei:
        eng.proc_begin()
    assert ei.value.code == -5                                      # MGN_E_UNSUPPORTED
    eng.close()
    # a mesh of the size the persistent kernels serve in the default mode (scattered labels on top: the mode goes through own_gid too)
    from util import scatter_labels
    pos, s, r = synth.mesh_1m(1234, 150, 150)
    pos, s, r, _ = scatter_labels(pos, s, r, 4)
    N, E = pos.shape[0], s.size
    rng = np.random.default_rng(2)
    for hl, mode in ((2, 0), (3, 1)):
        cfg = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=hl, mps=2)
        ps = orc.init_params(9, 3, 2, 128, hl, 2, seed=77 + hl, ln_jitter=0.1)
        nf, ef = rng.standard_normal((N, 9)).astype(np.float32), rng.standard_normal((E, 3)).astype(np.float32)
        v, e = rng.standard_normal((N, 128)).astype(np.float32), rng.standard_normal((E, 128)).astype(np.float32)
        orc.LN_MODE = mode
        try:
            ref_out = orc.forward(ps, cfg, nf, ef, s, r)
            rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 2)
        finally:
            orc.LN_MODE = 0
        eng = mgn_amd_engine(cfg, ln_dims="all", ln_mode=mode)
        eng.set_params(ps)
        eng.set_graph(s, r, N)
        assert rel_max(eng.forward(nf, ef), ref_out) <= TOL_15, (hl, mode)
        v1, e1 = eng.processor_steps(v, e, 2)
        assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15, (hl, mode)
        eng.close()


def mgn_amd_engine(cfg, **kw):
    import mgn_amd
    return mgn_amd.Engine(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], cfg["hidden_layers"], cfg["mps"], **kw)


@pytest.mark.parametrize("mode", [0, 1], ids=["var_eps", "std_eps"])
def test_training_under_whole_array_layernorm(whole_array_oracle, mode, monkeypatch):
    """mgn_step, mgn_forward_vjp and mgn_ode_vjp under ln_dims = MGN_LN_ALL: the pullback's two means run over the whole array (two
    column reductions and an elementwise pass around an MLP backward without LayerNorm).  Against the oracle's reverse mode, which is
    checked against finite differences under LN_DIMS = "all" on the CPU (tests/test_oracle_golden.py).  Cooperative kernels with graph
    replay (nx = 8), L = 32, the streaming kernels with a factored first layer (nx = 40, recompute on and off), hidden_layers = 3."""
    orc.LN_MODE = mode
    try:
        for L, nx, hl, recompute in ((128, 8, 2, None), (32, 8, 2, None), (128, 40, 2, "0"), (128, 40, 2, "1"), (128, 12, 3, None)):
            if recompute is None:
                monkeypatch.delenv("MGN_TRAIN_RECOMPUTE", raising=False)
            else:
                monkeypatch.setenv("MGN_TRAIN_RECOMPUTE", recompute)
            cfg = dict(Fn=9, Fe=3, O=2, L=L, hidden_layers=hl, mps=2)
            pos, s, r = small_mesh(nx, nx - 2)
            N, E = pos.shape[0], s.size
            ps = orc.init_params(9, 3, 2, L, hl, 2, seed=31 + L + hl, ln_jitter=0.1)
            nf, ef = random_inputs(N, E, cfg, 5)
            rng = np.random.default_rng(L + nx)
            target = rng.standard_normal((N, 2)).astype(np.float32)
            mask = rng.choice(N, N // 2, replace=False).astype(np.int32)
            eng = mgn_amd_engine(cfg, ln_dims="all", ln_mode=mode)
            eng.set_params(ps)
            eng.set_graph(s, r, N)
            g_ref, loss_ref = orc.step_grads(ps, cfg, nf, ef, s, r, target, mask)
            for rep in range(3):                                     # eager, capture, replay on the small meshes
                gs, loss = eng.step(nf, ef, target, mask)
                assert abs(loss - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref)), (L, nx, hl, rep)
                err = np.linalg.norm(gs - g_ref) / np.linalg.norm(g_ref)
                assert err <= 2e-4, (L, nx, hl, recompute, rep, err)
            # it is the other network: the row-wise LayerNorm's gradient is far away
            orc.LN_DIMS = "row"
            g_row, _ = orc.step_grads(ps, cfg, nf, ef, s, r, target, mask)
            orc.LN_DIMS = "all"
            assert np.linalg.norm(g_row - g_ref) > 20 * np.linalg.norm(gs - g_ref)
            ybar = rng.standard_normal((N, 2)).astype(np.float32)
            nfbar, gps, out = eng.forward_vjp(nf, ef, ybar, want_out=True)
            out_ref, gp_ref, nfbar_ref = orc.model_vjp(ps, cfg, nf, ef, s, r, lambda o: ybar.astype(np.float64))
            assert rel_max(out, out_ref) <= TOL_15
            # (4e-4 / 5e-4 on the 1 520-node mesh with a random cotangent: every LayerNorm couples all rows, rounding no longer stays in its row)
            assert np.linalg.norm(gps - gp_ref) <= 1e-3 * np.linalg.norm(gp_ref)
            assert np.linalg.norm(nfbar - nfbar_ref) <= 1e-3 * np.linalg.norm(nfbar_ref)
            if L == 128 and nx == 8:                                 # the right-hand side's pullback (solver-based training)
                x = (rng.standard_normal((N, 2)) * 0.3 + 1.0).astype(np.float32)
                onehot = np.eye(7, dtype=np.float32)[rng.integers(0, 7, N)]
                ef_raw = rng.standard_normal((E, 3)).astype(np.float32)
                vm = (rng.random(N) < 0.7).astype(np.float32)
                lam = rng.standard_normal((N, 2)).astype(np.float32)
                n_norm, t_norm = orc.NormMeanStd(np.array([1.0, 0.9]), np.array([0.31, 0.27])), orc.NormMinMax(0.0, 1.0)
                e_norm, o_norm = orc.NormMeanStd(ef_raw.mean(0), ef_raw.std(0)), orc.NormMeanStd(np.array([0.01, -0.02]), np.array([0.5, 0.4]))
                (ns, nsh), (ts, tsh), (es, esh) = n_norm.affine(2), t_norm.affine(7), e_norm.affine(3)
                eng.set_norms(node=(np.concatenate([ns, ts]), np.concatenate([nsh, tsh])), edge=(es, esh), out=(o_norm.std, o_norm.mean))
                xbar, gso, dxdt = eng.ode_vjp(x, onehot, ef_raw, lam, val_mask=vm, want_dxdt=True)
                rx, rg, rf = orc.ode_vjp(ps, cfg, x, onehot, ef_raw, s, r, n_norm, t_norm, e_norm, o_norm, vm, lam)
                assert rel_max(dxdt, rf) <= TOL_15
                assert np.linalg.norm(xbar - rx) <= 1e-3 * np.linalg.norm(rx) and np.linalg.norm(gso - rg) <= 2e-4 * np.linalg.norm(rg)
            eng.close()
    finally:
        orc.LN_MODE = 0


@pytest.mark.parametrize("N,E,seed", [(5, 0, 0), (33, 31, 2), (40, 700, 3), (70, 2049, 5)])
def test_whole_array_layernorm_on_ragged_graphs(whole_array_oracle, N, E, seed):
    """no edges at all (an empty array has no statistics and nothing to normalise), isolated nodes, self loops, duplicate edges, a heavy
    receiver: forward, processor steps, the right-hand side through the rollout driver, and the training step"""
    import warnings
    cfg = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=2, mps=2)
    s, r = synth.random_graph(N, E, seed)
    if E > 1000:
        r[:1000] = 3
    ps = orc.init_params(9, 3, 2, 128, 2, 2, seed=11 + seed, ln_jitter=0.1)
    rng = np.random.default_rng(seed)
    nf = rng.standard_normal((N, 9)).astype(np.float32)
    ef = rng.standard_normal((E, 3)).astype(np.float32)
    target = rng.standard_normal((N, 2)).astype(np.float32)
    mask = np.arange(0, N, 2, dtype=np.int32)
    eng = mgn_amd_engine(cfg, ln_dims="all")
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                              # (numpy's mean of an empty slice, E = 0)
        ref = orc.forward(ps, cfg, nf, ef, s, r)
        g_ref, loss_ref = orc.step_grads(ps, cfg, nf, ef, s, r, target, mask)
    assert rel_max(eng.forward(nf, ef), ref) <= TOL_15
    gs, loss = eng.step(nf, ef, target, mask)
    assert np.isfinite(gs).all() and abs(loss - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))
    assert np.linalg.norm(gs - g_ref) <= 1e-3 * np.linalg.norm(g_ref)
    # Euler through the native driver == repeated right-hand sides (identity normalisers)
    onehot = np.eye(7, dtype=np.float32)[rng.integers(0, 7, N)]
    x = rng.standard_normal((N, 2)).astype(np.float32)
    sol, _ = eng.rollout("Euler", x, onehot, ef, 0.0, 0.02, 0.01, 3, dt=0.01)
    xs = x.astype(np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(2):
            xs = xs + 0.01 * orc.forward(ps, cfg, np.concatenate([xs, onehot], 1), ef, s, r)
    assert rel_max(sol[2], xs) <= 1e-4
    eng.close()
