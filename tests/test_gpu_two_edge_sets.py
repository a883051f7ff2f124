"""Two edge sets (mesh + world edges; MGN-spec "per edge set", BASELINE.json configs[2] / GOLD-C): the HIP path through
the C ABI against the float64 oracle and the committed fixture.  The reference's FeatureGraph has one edge set
(src/graph.jl:87-96); this is the extension SURVEY.md 8c defines.  Run on the MI355X box with `-m gpu`."""
import os
from importlib import import_module

import numpy as np
import pytest
import torch   # before the first HIP call of libmgn_hip: torch must initialise its own runtime first

import mgn_amd
import mgn_oracle as orc
from mgn_amd import synth
from util import TOL_15, TOL_STEP, rel_max, set_kernel_path

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gold_c_two_sets.npz")
TOL_BF16 = 3e-2   # relative L2 after 15 steps


def rel_l2(a, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.linalg.norm(np.asarray(a, np.float64) - ref) / max(np.linalg.norm(ref), 1e-30))


def cfg2(L=128, mps=15, Fn=12, Fe=7, O=3, Fe2=4):
    return dict(Fn=Fn, Fe=Fe, O=O, L=L, hidden_layers=2, mps=mps, Fe2=Fe2)


def params2(cfg, seed=1234, jitter=0.1):
    return orc.init_params(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], 2, cfg["mps"], seed, jitter, Fe2=cfg["Fe2"])


def engine2(cfg, **kw):
    return mgn_amd.Engine(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], 2, cfg["mps"], Fe2=cfg["Fe2"], **kw)


@pytest.fixture(params=[1, 2, 3, 5], ids=["resident", "streaming", "cooperative", "cooperative16"])
def kernel_path(request):
    old = set_kernel_path(request.param)
    yield request.param
    set_kernel_path(old)


def test_param_count_matches_oracle_layout():
    for L, mps in ((32, 1), (128, 15)):
        cfg = cfg2(L, mps)
        eng = engine2(cfg)
        assert eng.param_count == orc.param_count(12, 7, 3, L, 2, mps, Fe2=4)
        ps = params2(cfg)
        eng.set_params(ps)
        assert np.array_equal(eng.get_params(), ps)


def test_gold_c_forward(kernel_path):
    g = np.load(GOLD)
    cfg = cfg2(int(g["L"]), int(g["mps"]))
    ps = params2(cfg, seed=int(g["seed"]), jitter=float(g["jitter"]))
    N = g["nf"].shape[0]
    eng = engine2(cfg)
    eng.set_params(ps)
    eng.set_graph(g["senders"], g["receivers"], N)
    eng.set_edge_set(1, g["senders2"], g["receivers2"])
    eng.set_edge_features(1, g["ef2"])
    out = eng.forward(g["nf"], g["ef"])
    assert rel_max(out, g["out"]) <= TOL_15, rel_max(out, g["out"])


def test_gold_c_latents_after_one_step(kernel_path):
    """Encoder + one processor step, every latent array (node, mesh edges, world edges) against the fixture."""
    g = np.load(GOLD)
    cfg = cfg2(int(g["L"]), int(g["mps"]))
    ps = params2(cfg, seed=int(g["seed"]), jitter=float(g["jitter"]))
    N = g["nf"].shape[0]
    eng = engine2(cfg)
    eng.set_params(ps)
    eng.set_graph(g["senders"], g["receivers"], N)
    eng.set_edge_set(1, g["senders2"], g["receivers2"])
    eng.set_edge_features(1, g["ef2"])
    eng.fwd_upload(g["nf"], g["ef"])
    eng.fwd_encode()
    eng.proc_edge(0)
    eng.proc_node(0, True)
    v, e = eng.latents_export()
    e2 = eng.edge_latents_export(1)
    assert rel_max(v, g["v_after_1"]) <= TOL_STEP * 2, rel_max(v, g["v_after_1"])   # encoder + step
    assert rel_max(e, g["e_after_1"]) <= TOL_STEP * 2
    assert rel_max(e2, g["e2_after_1"]) <= TOL_STEP * 2


@pytest.mark.parametrize("path", [0, 1], ids=["auto: 16-row kernels on bf16 storage", "bf16 MFMA kernels"])
def test_gold_c_bf16_band(path):
    old = set_kernel_path(path)
    try:
        g = np.load(GOLD)
        cfg = cfg2(int(g["L"]), int(g["mps"]))
        ps = params2(cfg, seed=int(g["seed"]), jitter=float(g["jitter"]))
        N = g["nf"].shape[0]
        eng = engine2(cfg, dtype="bf16")
        eng.set_params(ps)
        eng.set_graph(g["senders"], g["receivers"], N)
        eng.set_edge_set(1, g["senders2"], g["receivers2"])
        eng.set_edge_features(1, g["ef2"])
        out = eng.forward(g["nf"], g["ef"])
        assert rel_l2(out, g["out"]) <= TOL_BF16, rel_l2(out, g["out"])
    finally:
        set_kernel_path(old)


@pytest.mark.parametrize("L", [128, 64, 32])
def test_processor_steps_flag_mesh(L, kernel_path):
    """M-flag at BASELINE.json configs[2] size (40 x 40 cloth, ~9.3k mesh + ~12k world edges), 3 steps on given latents."""
    if L != 128 and kernel_path != 1:
        pytest.skip("kernel families only differ at L = 128")
    m = synth.mesh_flag()
    N, E, E2 = m["mesh_pos"].shape[0], m["s"].size, m["s2"].size
    assert N == 1600 and E2 > 1000
    cfg = cfg2(L, 3)
    ps = params2(cfg, jitter=0.05)
    rng = np.random.default_rng(11)
    v = rng.standard_normal((N, L)).astype(np.float32)
    e = rng.standard_normal((E, L)).astype(np.float32)
    e2 = rng.standard_normal((E2, L)).astype(np.float32)
    eng = engine2(cfg)
    eng.set_params(ps)
    eng.set_graph(m["s"], m["r"], N)
    eng.set_edge_set(1, m["s2"], m["r2"])
    eng.latents_import(v, e)
    eng.edge_latents_import(1, e2)
    eng.processor_steps_dev(3)
    v1, e1 = eng.latents_export()
    e21 = eng.edge_latents_export(1)
    rv, re, re2 = orc.processor_steps(ps, cfg, v, e, m["s"], m["r"], 3, set2=(e2, m["s2"], m["r2"]))
    assert rel_max(v1, rv) <= TOL_15, rel_max(v1, rv)
    assert rel_max(e1, re) <= TOL_15 and rel_max(e21, re2) <= TOL_15


def test_flag_mesh_15_steps_bf16_and_graph_replay():
    """configs[2] proper: M-flag, 15 steps, bf16; the second and third calls replay the captured hipGraph."""
    m = synth.mesh_flag()
    N, E, E2 = m["mesh_pos"].shape[0], m["s"].size, m["s2"].size
    cfg = cfg2(128, 15)
    ps = params2(cfg, jitter=0.05)
    rng = np.random.default_rng(12)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    e2 = rng.standard_normal((E2, 128)).astype(np.float32)
    rv, re, re2 = orc.processor_steps(ps, cfg, v, e, m["s"], m["r"], 15, set2=(e2, m["s2"], m["r2"]))
    for dtype, tol, path in (("f32", None, 0), ("bf16", TOL_BF16, 0), ("bf16", TOL_BF16, 1)):   # bf16: 16-row kernels on bf16 storage / bf16-MFMA kernels
        old_path = set_kernel_path(path)
        eng = engine2(cfg, dtype=dtype)
        eng.set_params(ps)
        eng.set_graph(m["s"], m["r"], N)
        eng.set_edge_set(1, m["s2"], m["r2"])
        for rep in range(3):
            eng.latents_import(v, e)
            eng.edge_latents_import(1, e2)
            eng.processor_steps_dev(15)
            v1, e1 = eng.latents_export()
            e21 = eng.edge_latents_export(1)
            if tol is None:
                assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15 and rel_max(e21, re2) <= TOL_15, rep
            else:
                assert rel_l2(v1, rv) <= tol and rel_l2(e1, re) <= tol and rel_l2(e21, re2) <= tol, (rep, rel_l2(v1, rv))
        set_kernel_path(old_path)


def test_world_edges_change_between_steps():
    """Cloth rollouts re-search the world edges every step: a handle whose second set is replaced must equal a fresh
    handle, and an EMPTY second set must equal the oracle with no world edges (aggregate of zeros)."""
    m = synth.mesh_flag(5, 14, 12, radius=0.12)
    N, E = m["mesh_pos"].shape[0], m["s"].size
    cfg = cfg2(128, 2)
    ps = params2(cfg)
    rng = np.random.default_rng(13)
    nf = rng.standard_normal((N, 12)).astype(np.float32)
    ef = rng.standard_normal((E, 7)).astype(np.float32)
    eng = engine2(cfg)
    eng.set_params(ps)
    eng.set_graph(m["s"], m["r"], N)
    keep = np.arange(m["s2"].size) % 3 != 0
    variants = [(m["s2"], m["r2"]), (m["s2"][keep], m["r2"][keep]), (m["s2"][:0], m["r2"][:0]), (m["s2"], m["r2"])]
    for s2, r2 in variants:
        ef2 = rng.standard_normal((s2.size, 4)).astype(np.float32)
        eng.set_edge_set(1, s2, r2)
        eng.set_edge_features(1, ef2)
        out = eng.forward(nf, ef)
        ref = orc.forward(ps, cfg, nf, ef, m["s"], m["r"], set2=(ef2, s2, r2))
        assert rel_max(out, ref) <= TOL_15, (s2.size, rel_max(out, ref))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("P", [2, 4])
def test_two_sets_partitioned_loopback(P, dtype):
    """KAT-7 for two edge sets: P edge-cut partitions on one GPU with the loopback halo exchange (rows carry the P row
    of both sets) against the single-partition oracle."""
    halo = import_module("mgn_amd.halo")
    m = synth.mesh_flag(7, 24, 20, radius=0.08)
    N, E, E2 = m["mesh_pos"].shape[0], m["s"].size, m["s2"].size
    cfg = cfg2(128, 3)
    ps = params2(cfg)
    rng = np.random.default_rng(14)
    v0 = rng.standard_normal((N, 128)).astype(np.float32)
    e0 = rng.standard_normal((E, 128)).astype(np.float32)
    e20 = rng.standard_normal((E2, 128)).astype(np.float32)
    stream = torch.cuda.current_stream().cuda_stream
    engs = []
    for k in range(P):
        g = engine2(cfg, rank=k, nranks=P, dtype=dtype)
        g.set_stream(stream)
        g.set_params(ps)
        g.set_graph(m["s"], m["r"], N, mesh_pos=m["mesh_pos"])
        g.set_edge_set(1, m["s2"], m["r2"])
        g.latents_import(v0, e0)
        g.edge_latents_import(1, e20)
        engs.append(g)
    assert sum(g.n_own for g in engs) == N and sum(g.e_local for g in engs) == E
    assert sum(g.edge_set_info(1)[1] for g in engs) == E2
    # the fold makes far-apart mesh regions world-neighbours: the halo is the union over both sets
    mgn_amd.run_processor_staged(engs, halo.LoopbackExchange(engs, torch.device("cuda")), 3)
    torch.cuda.synchronize()
    v, e, e2 = np.zeros((N, 128), np.float32), np.zeros((E, 128), np.float32), np.zeros((E2, 128), np.float32)
    for g in engs:
        g.latents_export(v, e)
        g.edge_latents_export(1, e2)
    rv, re, re2 = orc.processor_steps(ps, cfg, v0, e0, m["s"], m["r"], 3, set2=(e20, m["s2"], m["r2"]))
    if dtype == "bf16":
        assert rel_l2(v, rv) <= TOL_BF16 and rel_l2(e, re) <= TOL_BF16 and rel_l2(e2, re2) <= TOL_BF16
    else:
        assert rel_max(v, rv) <= TOL_15 and rel_max(e, re) <= TOL_15 and rel_max(e2, re2) <= TOL_15


def test_error_behaviour():
    cfg = cfg2(32, 1)
    one = mgn_amd.Engine(9, 3, 2, 32, 2, 1)
    one.set_graph(np.array([0, 1], np.int32), np.array([1, 0], np.int32), 2)
    with pytest.raises(mgn_amd.MgnError) as ei:
        one.set_edge_set(1, np.array([0], np.int32), np.array([1], np.int32))
    assert ei.value.code == -1
    two = engine2(cfg)
    with pytest.raises(mgn_amd.MgnError) as ei:
        two.set_edge_set(1, np.array([0], np.int32), np.array([1], np.int32))       # before set_graph
    assert ei.value.code == -3
    two.set_params(params2(cfg))
    two.set_graph(np.array([0, 1], np.int32), np.array([1, 0], np.int32), 2)
    two.set_edge_set(1, np.array([0], np.int32), np.array([1], np.int32))
    with pytest.raises(mgn_amd.MgnError) as ei:
        two.forward(np.zeros((2, 12), np.float32), np.zeros((2, 7), np.float32))    # world-edge features missing
    assert ei.value.code == -3
    with pytest.raises(mgn_amd.MgnError) as ei:                                     # single-edge-set RHS only
        two.ode_step(np.zeros((2, 3), np.float32), np.zeros((2, 9), np.float32), np.zeros((2, 7), np.float32))
    assert ei.value.code == -3
    with pytest.raises(mgn_amd.MgnError) as ei:
        two.set_edge_set(1, np.array([0], np.int32), np.array([5], np.int32))       # index out of range
    assert ei.value.code == -1
