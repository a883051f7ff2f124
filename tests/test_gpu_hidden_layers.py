"""`hidden_layers` other than 2 (reference Args.hidden_layers, src/MeshGraphNets.jl:35-38: any integer; MGN-spec: h hidden layers =
h + 1 Dense per MLP).  The tuned kernel families are specialised for the example's h = 2; every other count runs the GEN
instantiations (weights streamed from L2, a runtime loop over the middle layers).  Checked against the float64 oracle, against the
GOLD-F fixture, and -- at h = 2, where both exist -- against the tuned kernels."""
import os

import numpy as np
import pytest
import torch   # noqa: F401

import mgn_amd
import mgn_oracle as orc
from mgn_amd import synth
from mgn_amd.engine import MgnError
from util import TOL_15, engine_for, random_inputs, rel_max, set_fp32_split, set_kernel_path, small_mesh

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cfg_h(hl, L=128, mps=3, **kw):
    return dict(Fn=9, Fe=3, O=2, L=L, hidden_layers=hl, mps=mps, **kw)


@pytest.mark.parametrize("hl", [1, 3])
def test_gold_f_fixture(hl):
    g = np.load(os.path.join(GOLD, "gold_f_hidden_layers.npz"))
    cfg = cfg_h(hl, L=int(g["L"]), mps=int(g["mps"]))
    ps = orc.init_params(9, 3, 2, cfg["L"], hl, cfg["mps"], seed=int(g["seed"]) + hl, ln_jitter=float(g["jitter"]))
    assert ps.size == int(g[f"h{hl}_param_count"])
    eng = engine_for(cfg)
    assert eng.param_count == ps.size
    eng.set_params(ps)
    eng.set_graph(g["senders"], g["receivers"], g["nf"].shape[0])
    out = eng.forward(g["nf"], g["ef"])
    assert rel_max(out, g[f"h{hl}_out"]) <= TOL_15


@pytest.mark.parametrize("L", [128, 64, 32])
@pytest.mark.parametrize("hl", [1, 3, 4])
def test_forward_and_processor_against_the_oracle(hl, L):
    cfg = cfg_h(hl, L=L)
    pos, s, r = small_mesh(13, 9)
    N, E = pos.shape[0], s.size
    ps = orc.init_params(9, 3, 2, L, hl, 3, seed=5 + hl, ln_jitter=0.1)
    nf, ef = random_inputs(N, E, cfg, 1)
    eng = engine_for(cfg)
    eng.set_params(ps)
    assert np.array_equal(eng.get_params(), ps)
    eng.set_graph(s, r, N)
    assert rel_max(eng.forward(nf, ef), orc.forward(ps, cfg, nf, ef, s, r)) <= TOL_15
    rng = np.random.default_rng(2)
    v = rng.standard_normal((N, L)).astype(np.float32)
    e = rng.standard_normal((E, L)).astype(np.float32)
    v1, e1 = eng.processor_steps(v, e, 3)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 3)
    assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15


def test_gen_kernels_at_h2_equal_the_tuned_families():
    """Kernel path 4 forces the GEN instantiations at hidden_layers = 2: same chunk order, same summation order as the
    all-streaming tuned kernels -> the same bits.  Compared on the fp32-MFMA arithmetic both families share (the tuned encoders and
    decoder otherwise run on two fp16 pieces, which the GEN instantiations do not have); the default arithmetic is held to the oracle."""
    cfg = cfg_h(2, mps=4)
    pos, cells, _, _ = synth.mesh_cyl(7, 500)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = orc.init_params(9, 3, 2, 128, 2, 4, seed=3, ln_jitter=0.1)
    nf, ef = random_inputs(N, E, cfg, 4)
    outs = {}
    for path, split in ((2, 0), (4, 0), (2, 1)):
        old, old_split = set_kernel_path(path), set_fp32_split(split)
        try:
            eng = engine_for(cfg)
            eng.set_params(ps)
            eng.set_graph(s, r, N)
            outs[path, split] = eng.forward(nf, ef)
            eng.close()
        finally:
            set_kernel_path(old)
            set_fp32_split(old_split)
    assert np.array_equal(outs[2, 0], outs[4, 0])
    ref = orc.forward(ps, cfg, nf, ef, s, r)
    assert rel_max(outs[4, 0], ref) <= TOL_15 and rel_max(outs[2, 1], ref) <= TOL_15


def test_ragged_graph_rollout_and_two_edge_sets_with_h3():
    # ragged: a hub receiver spanning many edge tiles, isolated nodes
    cfg = cfg_h(3, mps=2)
    s, r = synth.random_graph(70, 2049, 5)
    r[:1000] = 3
    ps = orc.init_params(9, 3, 2, 128, 3, 2, seed=9, ln_jitter=0.1)
    nf, ef = random_inputs(70, 2049, cfg, 5)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, 70)
    assert rel_max(eng.forward(nf, ef), orc.forward(ps, cfg, nf, ef, s, r)) <= TOL_15
    # the fused right-hand side (normalisers, decoder epilogue) through the native Euler driver == repeated ode_step
    onehot = np.eye(7, dtype=np.float32)[np.random.default_rng(0).integers(0, 7, 70)]
    x = np.random.default_rng(1).standard_normal((70, 2)).astype(np.float32)
    eng.set_norms(node=(np.full(9, 0.5, np.float32), np.zeros(9, np.float32)), out=(np.full(2, 0.1, np.float32), np.zeros(2, np.float32)))
    sol, st = eng.rollout("Euler", x, onehot, ef, 0.0, 0.03, 0.01, 4, dt=0.01)
    xs = x.copy()
    for _ in range(3):
        xs = xs + np.float32(0.01) * eng.ode_step(xs, onehot, ef)
    assert st["n_rhs"] == 3 and rel_max(sol[3], xs) <= 1e-5
    eng.close()
    # two edge sets (flag_simple-shaped), h = 3
    mf = synth.mesh_flag(1234, 14, 12, radius=0.13)
    N = mf["mesh_pos"].shape[0]
    cfg2 = dict(Fn=12, Fe=7, O=3, L=128, hidden_layers=3, mps=2, Fe2=4)
    ps2 = orc.init_params(12, 7, 3, 128, 3, 2, 5, 0.1, Fe2=4)
    rng = np.random.default_rng(4)
    nf2 = rng.standard_normal((N, 12)).astype(np.float32)
    e2 = mgn_amd.Engine(12, 7, 3, 128, 3, 2, Fe2=4)
    e2.set_params(ps2)
    e2.set_graph(mf["s"], mf["r"], N)
    e2.set_edge_set(1, mf["s2"], mf["r2"])
    e2.set_edge_features(1, mf["ef2"])
    out = e2.forward(nf2, mf["ef"])
    ref = orc.forward(ps2, cfg2, nf2, mf["ef"], mf["s"], mf["r"], set2=(mf["ef2"], mf["s2"], mf["r2"]))
    assert rel_max(out, ref) <= TOL_15


def test_partitioned_h3_and_unsupported_combinations():
    import threading
    cfg = cfg_h(3, mps=3)
    pos, cells = synth.grid_mesh(21, 19, 2)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = orc.init_params(9, 3, 2, 128, 3, 3, seed=2, ln_jitter=0.1)
    nf, ef = random_inputs(N, E, cfg, 2)
    ref = orc.forward(ps, cfg, nf, ef, s, r)
    cid = mgn_amd.Engine.comm_unique_id("host")
    outs, errs = {}, []

    def body(k):
        try:
            e = engine_for(cfg, rank=k, nranks=2, device=0)
            e.set_params(ps)
            e.set_graph(s, r, N, mesh_pos=pos)
            e.comm_init(cid, "host")
            outs[k] = e.forward(nf, ef)
            e.comm_barrier()
            e.close()
        except BaseException as ex:   # noqa: BLE001
            errs.append(ex)

    ts = [threading.Thread(target=body, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    assert np.array_equal(outs[0], outs[1]) and rel_max(outs[0], ref) <= TOL_15
    # bf16 is specialised for hidden_layers = 2: refused, not silently wrong (the training step follows hidden_layers:
    # tests/test_gpu_training_general.py)
    with pytest.raises(MgnError):
        engine_for(cfg, dtype="bf16")
    for bad in (0, 5):
        with pytest.raises(MgnError):
            engine_for(cfg_h(bad))
