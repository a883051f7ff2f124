"""CPU tests: the float64 oracle against the committed golden vectors and known-answer tests; the fp32 C
restatement (oracle/mgn_ref.c) against the oracle; the reference-named host functions against both."""
import hashlib
import os

import numpy as np
import pytest

import mgn_oracle as orc
from util import rel_max

import mgn_amd
from importlib import import_module

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def params_for(g):
    ps = orc.init_params(9, 3, 2, int(g["L"]), 2, int(g["mps"]), seed=int(g["seed"]), ln_jitter=float(g["jitter"]))
    assert hashlib.sha256(ps.tobytes()).hexdigest() == str(g["params_sha256"]), "parameter generator drifted"
    return ps


@pytest.mark.parametrize("name,snaps", [("gold_a_L32_mps1.npz", [1]), ("gold_b_L128_mps15.npz", [1, 8, 15])])
def test_oracle_reproduces_golden(name, snaps):
    g = load(name)
    ps = params_for(g)
    cfg = dict(Fn=9, Fe=3, O=2, L=int(g["L"]), hidden_layers=2, mps=int(g["mps"]))
    out, lat = orc.forward(ps, cfg, g["nf"], g["ef"], g["senders"], g["receivers"], return_latents=True)
    assert rel_max(out, g["out"]) < 1e-12
    for k in snaps:
        assert rel_max(lat[k][0], g[f"v_after_{k}"]) < 1e-6
        assert rel_max(lat[k][1], g[f"e_after_{k}"]) < 1e-6


def test_oracle_ln_variants_gold_g():
    """GOLD-G: the LayerNorm spec variants as fixtures.  The (std + eps) denominator is a ~1e-5 relative
    effect; whole-array statistics are a different model -- the fixture is there so that the day julia/spec_probe.jl names one of
    them, the expected numbers already exist."""
    g = load("gold_g_ln_variants.npz")
    ps = orc.init_params(9, 3, 2, int(g["L"]), 2, int(g["mps"]), seed=int(g["seed"]), ln_jitter=float(g["jitter"]))
    assert hashlib.sha256(ps.tobytes()).hexdigest() == str(g["params_sha256"])
    cfg = dict(Fn=9, Fe=3, O=2, L=int(g["L"]), hidden_layers=2, mps=int(g["mps"]))
    for name, mode, dims in (("out_v1", 0, "row"), ("out_std_eps", 1, "row"), ("out_whole_array", 0, "all")):
        orc.LN_MODE, orc.LN_DIMS = mode, dims
        try:
            out = orc.forward(ps, cfg, g["nf"], g["ef"], g["senders"], g["receivers"])
        finally:
            orc.LN_MODE, orc.LN_DIMS = 0, "row"
        assert rel_max(out, g[name]) < 1e-12, name
    d1 = rel_max(g["out_std_eps"], g["out_v1"])
    d2 = rel_max(g["out_whole_array"], g["out_v1"])
    assert 1e-7 < d1 < 1e-3 and d2 > 1e-2, (d1, d2)


def test_oracle_reproduces_gold_c_two_edge_sets():
    """GOLD-C: two edge sets (flag_simple-shaped widths).  Also pins that an EMPTY second set only contributes its
    zero aggregate (the node MLP still has the wider first layer)."""
    g = load("gold_c_two_sets.npz")
    cfg = dict(Fn=12, Fe=7, O=3, L=int(g["L"]), hidden_layers=2, mps=int(g["mps"]), Fe2=4)
    ps = orc.init_params(12, 7, 3, cfg["L"], 2, cfg["mps"], seed=int(g["seed"]), ln_jitter=float(g["jitter"]), Fe2=4)
    assert hashlib.sha256(ps.tobytes()).hexdigest() == str(g["params_sha256"]), "parameter generator drifted"
    assert ps.size == orc.param_count(12, 7, 3, cfg["L"], 2, cfg["mps"], Fe2=4)
    set2 = (g["ef2"], g["senders2"], g["receivers2"])
    out, lat = orc.forward(ps, cfg, g["nf"], g["ef"], g["senders"], g["receivers"], return_latents=True, set2=set2)
    assert rel_max(out, g["out"]) < 1e-12
    assert rel_max(lat[1][0], g["v_after_1"]) < 1e-6 and rel_max(lat[1][1], g["e_after_1"]) < 1e-6
    assert rel_max(lat[1][2], g["e2_after_1"]) < 1e-6 and rel_max(lat[15][0], g["v_after_15"]) < 1e-6
    # world edges matter: dropping them changes the answer
    empty = (g["ef2"][:0], g["senders2"][:0], g["receivers2"][:0])
    out0 = orc.forward(ps, cfg, g["nf"], g["ef"], g["senders"], g["receivers"], set2=empty)
    assert rel_max(out0, g["out"]) > 1e-3


@pytest.fixture(params=[(0, "row"), (1, "row"), (0, "all"), (1, "all")], ids=["ln_var_eps", "ln_std_eps", "ln_all_var_eps", "ln_all_std_eps"])
def ln_mode(request):
    """both LayerNorm denominators and both statistics ranges (DESIGN.md spec_variant): the reverse mode follows orc.LN_MODE / orc.LN_DIMS"""
    orc.LN_MODE, orc.LN_DIMS = request.param
    yield request.param
    orc.LN_MODE, orc.LN_DIMS = 0, "row"


def test_step_grads_match_finite_differences(ln_mode):
    """The hand-written reverse mode of the oracle (the checker of mgn_step) against central differences of its own
    float64 forward: loss = mean(mse_reduce(target, model(graph))[mask]) (reference src/strategies.jl:418-422)."""
    cfg = dict(Fn=9, Fe=3, O=2, L=32, hidden_layers=2, mps=2)
    ps = orc.init_params(9, 3, 2, 32, 2, 2, seed=1, ln_jitter=0.1).astype(np.float64)
    pos, cells = mgn_amd.synth.grid_mesh(5, 4, 3)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    rng = np.random.default_rng(0)
    nf, ef, tgt = rng.standard_normal((N, 9)), rng.standard_normal((E, 3)), rng.standard_normal((N, 2))
    mask = np.array([0, 3, 4, 7, 11, 12, 12])                     # a node listed twice counts twice, like err[mask]
    g, loss = orc.step_grads(ps, cfg, nf, ef, s, r, tgt, mask)
    assert abs(loss - orc.loss_only(ps, cfg, nf, ef, s, r, tgt, mask)) < 1e-12
    assert g.size == ps.size
    # one probe in every parameter tensor
    off = 0
    for bname, tensors in orc.model_layout(9, 3, 2, 32, 2, 2):
        for tname, shape in tensors:
            n = int(np.prod(shape))
            i = off + int(rng.integers(n))
            hh = 1e-6
            p1, p2 = ps.copy(), ps.copy()
            p1[i] += hh
            p2[i] -= hh
            fd = (orc.loss_only(p1, cfg, nf, ef, s, r, tgt, mask) - orc.loss_only(p2, cfg, nf, ef, s, r, tgt, mask)) / (2 * hh)
            assert abs(fd - g[i]) <= 1e-6 * max(abs(fd), 1.0), (bname, tname, fd, g[i])
            off += n


@pytest.mark.parametrize("hidden_layers,two_sets", [(1, False), (3, False), (4, False), (2, True), (3, True)])
def test_step_grads_match_finite_differences_general(hidden_layers, two_sets):
    """the same for hidden_layers != 2 and for the second edge set of MGN-spec (the checker of mgn_step's general path)"""
    Fe2 = 4 if two_sets else None
    cfg = dict(Fn=9, Fe=3, O=2, L=32, hidden_layers=hidden_layers, mps=2)
    if two_sets:
        cfg["Fe2"] = Fe2
    ps = orc.init_params(9, 3, 2, 32, hidden_layers, 2, seed=2, ln_jitter=0.1, Fe2=Fe2).astype(np.float64)
    pos, cells = mgn_amd.synth.grid_mesh(5, 4, 3)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    rng = np.random.default_rng(3)
    nf, ef, tgt = rng.standard_normal((N, 9)), rng.standard_normal((E, 3)), rng.standard_normal((N, 2))
    set2 = None
    if two_sets:
        s2 = rng.integers(0, N, 17).astype(np.int32)
        r2 = np.sort(rng.integers(0, N, 17)).astype(np.int32)
        set2 = (rng.standard_normal((17, Fe2)), s2, r2)
    mask = np.array([0, 3, 4, 7, 11, 12, 12])
    g, loss = orc.step_grads(ps, cfg, nf, ef, s, r, tgt, mask, set2=set2)
    assert abs(loss - orc.loss_only(ps, cfg, nf, ef, s, r, tgt, mask, set2=set2)) < 1e-12
    assert g.size == ps.size
    off = 0
    for bname, tensors in orc.model_layout(9, 3, 2, 32, hidden_layers, 2, Fe2):
        for tname, shape in tensors:
            n = int(np.prod(shape))
            i = off + int(rng.integers(n))
            hh = 1e-6
            p1, p2 = ps.copy(), ps.copy()
            p1[i] += hh
            p2[i] -= hh
            fd = (orc.loss_only(p1, cfg, nf, ef, s, r, tgt, mask, set2=set2) - orc.loss_only(p2, cfg, nf, ef, s, r, tgt, mask, set2=set2)) / (2 * hh)
            assert abs(fd - g[i]) <= 1e-6 * max(abs(fd), 1.0), (bname, tname, fd, g[i])
            off += n


def test_param_count_matches_survey():
    # enc-node 34,560; enc-edge 33,792; per step edge 82,560 + node 66,176; decoder 33,282
    assert orc.param_count(9, 3, 2, 128, 2, 15) == 34560 + 33792 + 15 * (82560 + 66176) + 33282


def test_kats():
    k = load("kats.npz")
    ref = import_module("mgn_amd.reference_api")
    s1, r1 = ref.triangles_to_edges(np.array([[0, 1, 2]]))
    assert s1.size == 6 and np.array_equal(s1, k["tri1_s"]) and np.array_equal(r1, k["tri1_r"])       # KAT-1
    s2, r2 = ref.triangles_to_edges(np.array([[0, 1, 2], [1, 3, 2]]))
    assert s2.size == 10 and np.array_equal(s2, k["tri2_s"]) and np.array_equal(r2, k["tri2_r"])
    assert np.array_equal(ref.one_hot(k["onehot_types"], 7, 0), k["onehot"].astype(np.float32))      # KAT-2
    data = dict(node_type=np.zeros(3, np.int32), mesh_pos=np.array([[0, 0], [3, 0], [3, 4]], np.float32),
                edges=np.array([[1, 0], [2, 1], [2, 0]]))
    _, s, r, ef = ref.create_base_graph(data, 6, 0)
    assert sorted(ef[:3, 2].tolist()) == [3.0, 4.0, 5.0]                                             # KAT-3
    assert np.allclose(ef[3:, :2], -ef[:3, :2])
    # vectorised mesh helper gives the same edge SET as the reference-order function
    pos, cells = mgn_amd.synth.grid_mesh(7, 5, 0)
    a = set(zip(*[x.tolist() for x in mgn_amd.synth.cells_to_edges(cells)]))
    b = set(zip(*[x.tolist() for x in ref.triangles_to_edges(cells)]))
    assert a == b


def test_kat5_scatter_star_and_ln_constant():
    rows = np.arange(12.0).reshape(4, 3)
    assert np.array_equal(orc.scatter_add(rows, np.zeros(4, int), 2)[0], rows.sum(0))                # KAT-5
    x = np.full((2, 8), 3.5)
    assert np.allclose(orc.layer_norm(x, np.ones(8), np.full(8, 0.25)), 0.25)                        # KAT-4 mechanism


def test_c_restatement_matches_oracle_on_golden():
    import mgn_ref
    for name in ("gold_a_L32_mps1.npz", "gold_b_L128_mps15.npz"):
        g = load(name)
        ps = params_for(g)
        cfg = dict(Fn=9, Fe=3, O=2, L=int(g["L"]), hidden_layers=2, mps=int(g["mps"]))
        out = mgn_ref.forward(ps, cfg, g["nf"], g["ef"], g["senders"], g["receivers"])
        assert rel_max(out, g["out"]) <= 1e-4, name
    g = load("gold_b_L128_mps15.npz")
    ps = params_for(g)
    cfg = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=2, mps=15)
    v, e = mgn_ref.processor_steps(ps, cfg, g["v_after_1"], g["e_after_1"], g["senders"], g["receivers"], 0)
    assert np.array_equal(v, g["v_after_1"])


def test_c_restatement_ragged():
    import mgn_ref
    cfg = dict(Fn=9, Fe=3, O=2, L=32, hidden_layers=2, mps=2)
    ps = orc.init_params(9, 3, 2, 32, 2, 2, 7, 0.1)
    for N, E, seed in [(1, 0, 0), (5, 1, 1), (70, 300, 2)]:
        s, r = mgn_amd.synth.random_graph(N, E, seed)
        rng = np.random.default_rng(seed)
        nf, ef = rng.standard_normal((N, 9)).astype(np.float32), rng.standard_normal((E, 3)).astype(np.float32)
        assert rel_max(mgn_ref.forward(ps, cfg, nf, ef, s, r), orc.forward(ps, cfg, nf, ef, s, r)) <= 1e-4


def test_normalisers_and_rollout_wrapper_match_oracle():
    """reference_api (float32 host mirror of src/graph.jl + src/solve.jl) against the oracle's ode_rhs on GOLD-D,
    with the model call served by the oracle (no GPU here)."""
    ref = import_module("mgn_amd.reference_api")
    g = load("gold_d_rollout.npz")
    ps = params_for(g)
    cfg = dict(Fn=9, Fe=3, O=2, L=int(g["L"]), hidden_layers=2, mps=int(g["mps"]))

    class FakeMgn:   # GraphNetwork-shaped holder whose model is the oracle
        def __init__(self):
            self.ps, self.st = ps, None
            self.n_norm = {"velocity": ref.NormaliserOfflineMeanStd(-g["node_shift"][:2] / g["node_scale"][:2], 1 / g["node_scale"][:2]),
                           "node_type": ref.NormaliserOfflineMinMax(0.0, 1.0)}
            self.e_norm = ref.NormaliserOfflineMeanStd(-g["edge_shift"] / g["edge_scale"], 1 / g["edge_scale"])
            self.o_norm = {"velocity": ref.NormaliserOfflineMeanStd(g["out_shift"], g["out_scale"])}

        def model(self, graph, ps_, st):
            return orc.forward(ps_, cfg, graph.nf, graph.ef, graph.senders, graph.receivers).astype(np.float32), st

    mgn = FakeMgn()
    data = dict(node_type=g["node_type"], mesh_pos=g["mesh_pos"], edges=np.stack([g["senders"], g["receivers"]], 1)[: g["senders"].size // 2])
    onehot, s, r, ef = ref.create_base_graph(data, 6, 0)
    assert np.array_equal(s, g["senders"]) and np.array_equal(r, g["receivers"])
    assert np.allclose(ef, g["ef_raw"], atol=1e-6)
    meta = {"features": {"velocity": {"dim": 2}}}
    val_mask = g["val_mask"][:, None].astype(np.float32)
    saves = np.arange(11) * float(g["dt"])
    sol_u, sol_t = ref.rollout("Euler", mgn, {"velocity": g["x0"].astype(np.float32)}, ["velocity"], meta, ["velocity"],
                               {"velocity": 2}, onehot, ef, s, r, val_mask, g["inflow_mask"], {"velocity": g["gt"]},
                               0.0, 10 * float(g["dt"]), float(g["dt"]), saves)
    assert sol_u.shape == g["xs"].shape
    assert np.linalg.norm(sol_u - g["xs"]) / np.linalg.norm(g["xs"]) <= 1e-5


def test_inflow_frame_rule_kat():
    """KAT-8: the inflow frame of a right-hand side is `floor(Int, t / saves_dt) + 1` in the solver's own time type with no tolerance
    (reference src/solve.jl:151).  In Float64 0.29 / 0.01 = 28.999999999999996 floors to 28 (1-based 29): the reference reads the frame
    of the PREVIOUS save point there (0.07 / 0.01 is 7.000000000000001 and floors to 7); in Float32 the quotient is exactly 29.
    The oracle restates the rule, the twin follows it."""
    assert orc.inflow_frame(0.29, 0.01, "reference", np.float64) == 28
    assert orc.inflow_frame(0.29, 0.01, "reference", np.float32) == 29
    assert orc.inflow_frame(0.29, 0.01, "tolerant", np.float64) == 29
    assert orc.inflow_frame(0.07, 0.01, "reference", np.float64) == 7
    # on the integrator's own times (t <- t + dt): first stale frame at step 6 in Float32, at step 10 in Float64
    for T, stale in ((np.float32, 6), (np.float64, 10)):
        fr = [orc.inflow_frame(t, 0.01, "reference", T) for t in orc.euler_times(0.0, 0.01, 16, T)]
        assert fr[:stale] == list(range(stale)) and fr[stale] == stale - 1
    # the twin of ode_func_eval applies the expression to whatever type t and saves_dt arrive in
    ref = import_module("mgn_amd.reference_api")
    seen = []

    class Mgn:
        ps = st = None
        n_norm = {"velocity": lambda x: x, "node_type": lambda x: x}
        e_norm = staticmethod(lambda x: x)
        o_norm = {"velocity": ref.NormaliserOfflineMeanStd(np.zeros(2, np.float32), np.ones(2, np.float32))}

        def model(self, graph, ps_, st):
            return np.zeros((graph.nf.shape[0], 2), np.float32), st

    data = {"velocity": np.arange(31, dtype=np.float32)[:, None, None] * np.ones((31, 3, 2), np.float32)}
    for T, t, want in ((np.float64, 0.29, 28), (np.float32, 0.29, 29)):
        x = np.zeros((3, 2), np.float32)
        p = (Mgn(), None, data, {}, ["velocity"], {"features": {"velocity": {"dim": 2}}}, ["velocity"], {"velocity": 2},
             np.zeros((3, 1), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.int32), np.zeros(0, np.int32),
             np.ones((3, 1), np.float32), np.ones((3, 2), bool), T(0.01), None)
        ref.ode_func_eval(x, p, T(t))
        assert x[0, 0] == want                                     # the rows were overwritten from frame `want`
    with pytest.raises(IndexError):                                # BoundsError in the reference
        ref.ode_func_eval(np.zeros((3, 2), np.float32), p, np.float32(0.5))


def test_online_normaliser_accumulates_then_freezes():
    ref = import_module("mgn_amd.reference_api")
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1000, 3)) * [1, 2, 3] + [5, 0, -1]).astype(np.float32)
    n = ref.NormaliserOnline(3, max_acc=2)
    n(x[:500]); n(x[500:])
    frozen = n.frozen()
    n(x[:10] * 100)   # third call: max_acc reached, statistics must not move
    assert np.allclose(n.frozen().mean, frozen.mean) and np.allclose(frozen.mean, x.mean(0), atol=1e-3)
    assert np.allclose(n.inverse(n(x, accumulate=False)), x, atol=1e-4)
    with pytest.raises(ValueError):
        n(x[:, :2])


def test_tsit5_tableau_and_convergence():
    """KAT for the adaptive branch of rollout: Tsitouras 5(4) order conditions and convergence on y' = -y + sin t."""
    A, c, bt = orc.TS_A, orc.TS_C, orc.TS_BT
    b = np.append(A[6], 0.0)
    assert np.allclose(A.sum(1), c, atol=1e-15) and abs(bt.sum()) < 1e-15
    for p, want in [(0, 1.0), (1, 1 / 2), (2, 1 / 3), (3, 1 / 4), (4, 1 / 5)]:
        assert abs(b @ c ** p - want) < 1e-14
    f = lambda x, t: -x + np.sin(t)
    exact = 1.5 * np.exp(-2.0) + 0.5 * (np.sin(2.0) - np.cos(2.0))
    errs = []
    for tol in (1e-4, 1e-6, 1e-8):
        sol, st = orc.tsit5_rollout(f, np.array([1.0]), 0.0, 2.0, np.linspace(0, 2, 5), abstol=tol * 1e-3, reltol=tol)
        errs.append(abs(sol[-1, 0] - exact))
        assert st["n_rhs"] == 2 + 6 * (st["n_accept"] + st["n_reject"])      # FSAL: 6 evaluations per step
    assert errs[0] > errs[1] > errs[2] and errs[2] < 1e-9
