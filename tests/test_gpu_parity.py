"""GPU parity tests proper: the HIP path (through the C ABI) against the float64 oracle on identical
seeded inputs.  Run on the MI355X box with `-m gpu`."""
import numpy as np
import pytest
import torch   # before the engine's first HIP call (device-array tests)

import mgn_oracle as orc
from util import (TOL_15, TOL_STEP, cfg_dict, engine_for, make_params, random_inputs, rel_max, small_mesh)

from mgn_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 2, 3, 5, (5, 2), (5, 3)], ids=["resident", "streaming", "cooperative", "cooperative16", "cooperative16x2", "cooperative16x3"])
def kernel_path(request):
    """every kernel family; (5, rt): the 16-row cooperative kernels with rt 16-edge tiles per block of the edge kernel"""
    from util import set_c16_row_tiles, set_kernel_path
    path, rt = request.param if isinstance(request.param, tuple) else (request.param, 0)
    old, old_rt = set_kernel_path(path), set_c16_row_tiles(rt)
    yield request.param
    set_kernel_path(old)
    set_c16_row_tiles(old_rt)


@pytest.mark.parametrize("L", [128, 64, 32])
def test_forward_small_mesh(L):
    cfg = cfg_dict(L=L, mps=3)
    pos, s, r = small_mesh()
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    nf, ef = random_inputs(N, E, cfg)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    out = eng.forward(nf, ef)
    ref = orc.forward(ps, cfg, nf, ef, s, r)
    assert rel_max(out, ref) <= TOL_15, rel_max(out, ref)


def test_processor_one_step_latents(kernel_path):
    cfg = cfg_dict(mps=1)
    pos, s, r = small_mesh(11, 7)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    rng = np.random.default_rng(5)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    v1, e1 = eng.processor_steps(v, e, 1)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 1)
    assert rel_max(e1, re) <= TOL_STEP, ("edge", rel_max(e1, re))
    assert rel_max(v1, rv) <= TOL_STEP, ("node", rel_max(v1, rv))


def test_processor_15_steps_cyl(kernel_path):
    """cfg-2 shaped: M-cyl, L=128, 15 steps (GOLD-B shape at full size)."""
    cfg = cfg_dict(mps=15)
    pos, cells, node_type, vel = synth.mesh_cyl(1234, 600)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg, jitter=0.05)
    rng = np.random.default_rng(7)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    v1, e1 = eng.processor_steps(v, e, 15)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 15)
    assert rel_max(e1, re) <= TOL_15, ("edge", rel_max(e1, re))
    assert rel_max(v1, rv) <= TOL_15, ("node", rel_max(v1, rv))


@pytest.mark.parametrize("N,E,seed", [(1, 0, 0), (5, 1, 1), (33, 31, 2), (40, 700, 3), (64, 64, 4), (70, 2049, 5)])
def test_ragged_graphs(N, E, seed, kernel_path):
    """Ragged inputs: isolated nodes, self loops, duplicate edges, receivers with > 32 and > 64
    incoming edges (segments that straddle one and several 32-edge tiles), E not a multiple of 32."""
    cfg = cfg_dict(mps=2)
    s, r = synth.random_graph(N, E, seed)
    if E >= 700:
        r[: E // 2] = 3      # one hub receiver spanning many tiles
        r[E // 2: E // 2 + 40] = 7
    ps = make_params(cfg)
    nf, ef = random_inputs(N, E, cfg, seed)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    out = eng.forward(nf, ef)
    ref = orc.forward(ps, cfg, nf, ef, s, r)
    assert rel_max(out, ref) <= TOL_15, rel_max(out, ref)


def test_edge_order_permutation_invariance():
    """KAT-6: the engine re-sorts edges by receiver, so any input edge order gives the same result
    bit for bit (stable sort keeps the within-receiver order only up to the permutation; tolerance
    is a few ulp of the aggregate)."""
    cfg = cfg_dict(mps=2)
    pos, s, r = small_mesh(9, 9)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    nf, ef = random_inputs(N, E, cfg, 9)
    perm = np.random.default_rng(0).permutation(E)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    a = eng.forward(nf, ef)
    eng.set_graph(s[perm], r[perm], N)
    b = eng.forward(nf, ef[perm])
    assert rel_max(b, a) <= 2e-6


def test_one_based_indices_and_determinism(kernel_path):
    cfg = cfg_dict(mps=2)
    pos, s, r = small_mesh()
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    nf, ef = random_inputs(N, E, cfg, 2)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N, index_base=0)
    a = eng.forward(nf, ef)
    eng.set_graph(s + 1, r + 1, N, index_base=1)   # Julia boundary, src/graph.jl:31-34
    b = eng.forward(nf, ef)
    c = eng.forward(nf, ef)
    assert np.array_equal(a, b) and np.array_equal(b, c)   # no atomics: bitwise reproducible


def test_kat4_zero_weights_give_decoder_bias():
    cfg = cfg_dict(mps=2)
    pos, s, r = small_mesh()
    N, E = pos.shape[0], s.size
    ps = np.zeros(orc.param_count(9, 3, 2, 128, 2, 2), np.float32)
    P = orc.unpack_params(ps, 9, 3, 2, 128, 2, 2)
    P["decoder"]["b3"][:] = [0.25, -1.5]
    nf, ef = random_inputs(N, E, cfg)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    out = eng.forward(nf, ef)
    assert np.allclose(out, np.array([0.25, -1.5], np.float32)[None, :], atol=0)


def test_error_paths():
    from mgn_amd import MgnError
    cfg = cfg_dict(mps=1)
    eng = engine_for(cfg)
    with pytest.raises(MgnError):
        eng.forward(np.zeros((0, 9), np.float32), np.zeros((0, 3), np.float32))   # before set_params/set_graph
    with pytest.raises(MgnError):
        eng.set_params(np.zeros(10, np.float32))
    eng.set_params(make_params(cfg))
    with pytest.raises(MgnError):
        eng.set_graph(np.array([0, 5], np.int32), np.array([1, 1], np.int32), 3)  # index out of range
    with pytest.raises(MgnError):
        engine_for(dict(cfg, L=100))


def test_mid_size_mesh_with_tail_split():
    """90 000 nodes / 537 602 edges (16.8 k edge tiles): the LDS-resident persistent kernels with the last partial round of
    the edge walk handed to the cooperative kernel -- two kernel families inside one edge step, against the oracle."""
    from mgn_amd import synth as sy
    pos, s, r = sy.mesh_1m(7, 300, 300)
    N, E = pos.shape[0], s.size
    cfg = cfg_dict(mps=2)
    ps = make_params(cfg, seed=3, jitter=0.05)
    rng = np.random.default_rng(0)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    v1, e1 = eng.processor_steps(v, e, 2)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 2)
    assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_degenerate_graphs_both_precisions(dtype):
    """One node, no edges, fewer edges than a tile, more edges than nodes -- and a two-edge-set model whose mesh set is
    empty while the world set is not."""
    import mgn_amd
    tol = 3e-2 if dtype == "bf16" else 1e-4
    l2 = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    cfg = cfg_dict(mps=2)
    ps = make_params(cfg, seed=1, jitter=0.05)
    for N, E in ((1, 0), (2, 0), (33, 1), (5, 40)):
        rng = np.random.default_rng(N * 7 + E)
        s, r = rng.integers(0, N, E).astype(np.int32), rng.integers(0, N, E).astype(np.int32)
        nf, ef = rng.standard_normal((N, 9)).astype(np.float32), rng.standard_normal((E, 3)).astype(np.float32)
        eng = engine_for(cfg, dtype=dtype)
        eng.set_params(ps)
        eng.set_graph(s, r, N)
        assert l2(eng.forward(nf, ef), orc.forward(ps, cfg, nf, ef, s, r)) <= tol, (N, E)
    cfg2 = dict(Fn=12, Fe=7, O=3, L=128, hidden_layers=2, mps=2, Fe2=4)
    ps2 = orc.init_params(12, 7, 3, 128, 2, 2, seed=2, ln_jitter=0.05, Fe2=4)
    rng = np.random.default_rng(5)
    N = 40
    s2, r2 = rng.integers(0, N, 70).astype(np.int32), rng.integers(0, N, 70).astype(np.int32)
    nf, ef2 = rng.standard_normal((N, 12)).astype(np.float32), rng.standard_normal((70, 4)).astype(np.float32)
    eng = mgn_amd.Engine(12, 7, 3, 128, 2, 2, dtype=dtype, Fe2=4)
    eng.set_params(ps2)
    none = np.zeros(0, np.int32)
    eng.set_graph(none, none, N)
    eng.set_edge_set(1, s2, r2)
    eng.set_edge_features(1, ef2)
    out = eng.forward(nf, np.zeros((0, 7), np.float32))
    ref = orc.forward(ps2, cfg2, nf, np.zeros((0, 7)), np.zeros(0, int), np.zeros(0, int), set2=(ef2, s2, r2))
    assert l2(out, ref) <= tol


def test_feature_stats_online_normaliser_accumulation():
    """mgn_feature_stats == one accumulation of GraphNetCore's NormaliserOnline: float64 column sums / sums of squares,
    host or device source, ragged row counts (block edges), bitwise repeatable; the Python mirror gives the same
    normalised output with the engine-backed accumulation."""
    import torch
    from mgn_amd import reference_api as ra
    eng = engine_for(cfg_dict(L=32, mps=1))
    rng = np.random.default_rng(5)
    for rows, dim in ((1, 3), (2047, 9), (2049, 12), (50000, 128), (7, 33)):
        x = (rng.standard_normal((rows, dim)) * 3 + 1).astype(np.float32)
        s, q = eng.feature_stats(x)
        x64 = x.astype(np.float64)
        assert np.allclose(s, x64.sum(0), rtol=1e-12, atol=1e-9) and np.allclose(q, (x64 ** 2).sum(0), rtol=1e-12, atol=1e-9)
        s2, q2 = eng.feature_stats(torch.from_numpy(x).cuda())
        assert np.array_equal(s, s2) and np.array_equal(q, q2)
    with pytest.raises(ValueError):
        eng.feature_stats(np.zeros(5, np.float32))
    a, b = ra.NormaliserOnline(9), ra.NormaliserOnline(9)
    b.engine = eng
    for _ in range(3):
        x = rng.standard_normal((4000, 9)).astype(np.float32) * 2 - 0.5
        ya, yb = a(x), b(x)
    assert np.allclose(ya, yb, rtol=1e-6, atol=1e-6)


def test_compute_from_plain_c(tmp_path):
    """The compute entry points driven from a C99 program (tests/c_abi/abi_gpu.c: no Python, no C++ between it and the library):
    mgn_forward and mgn_step on a small mesh, against the oracle and bitwise against the same calls through ctypes."""
    import os
    import subprocess
    import mgn_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(mgn_amd.lib_path()) if hasattr(mgn_amd, "lib_path") else os.path.join(root, "meshgraphnets.jl_amd", "lib")
    exe = str(tmp_path / "abi_gpu")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", os.path.join(root, "tests", "c_abi", "abi_gpu.c"),
                           "-o", exe, "-L", libdir, "-lmgn_hip", "-Wl,-rpath," + libdir])
    cfg = cfg_dict(mps=3)
    pos, s, r = small_mesh(11, 8)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg).astype(np.float32)
    nf, ef = random_inputs(N, E, cfg, 5)
    rng = np.random.default_rng(6)
    target = rng.standard_normal((N, 2)).astype(np.float32)
    mask = np.sort(rng.choice(N, N // 2, replace=False)).astype(np.int32)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([N, E, ps.size, cfg["mps"], mask.size], np.int32).tobytes())
        for a in (s.astype(np.int32), r.astype(np.int32), mask, ps, nf, ef, target):
            f.write(np.ascontiguousarray(a).tobytes())
    run = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "abi_gpu OK" in run.stdout, run.stdout + run.stderr
    raw = np.fromfile(tmp_path / "out.bin", np.float32)
    out_c, loss_c, gs_c = raw[:2 * N].reshape(N, 2), float(raw[2 * N]), raw[2 * N + 1:]
    assert gs_c.size == ps.size
    assert rel_max(out_c, orc.forward(ps, cfg, nf, ef, s, r)) <= TOL_15
    ref_g, ref_loss = orc.step_grads(ps, cfg, nf, ef, s, r, target, mask)
    assert abs(loss_c - ref_loss) <= 1e-5 * abs(ref_loss)
    assert np.linalg.norm(gs_c - ref_g) <= 1e-3 * np.linalg.norm(ref_g)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    assert np.array_equal(eng.forward(nf, ef), out_c)
    gs_py, loss_py = eng.step(nf, ef, target, mask)
    assert loss_py == loss_c and np.array_equal(gs_py, gs_c)


@pytest.mark.parametrize("dtype,L,hl,two", [("f32", 128, 2, False), ("f32", 128, 3, False), ("f32", 64, 2, False), ("bf16", 128, 2, True), ("f32", 128, 2, True)])
def test_device_packed_weight_layouts_equal_the_host_specification(dtype, L, hl, two):
    """the kernels' weight layouts are written on the device from the uploaded parameter vector (k_pack_weights); the host functions
    pack_chunk / _tmajor / 16 / _bf16 / 16_bf16 / _split are their specification: every chunk must be bitwise equal"""
    import ctypes
    import mgn_amd
    kw = dict(Fe2=4) if two else {}
    eng = mgn_amd.Engine(9, 3, 2, L, hl, 3, dtype=dtype, **kw)
    ps = orc.init_params(9, 3, 2, L, hl, 3, 77, 0.05, **kw)
    eng.set_params(ps)
    lib = mgn_amd.load()
    lib.mgn_debug_pack_check.restype = ctypes.c_longlong
    lib.mgn_debug_pack_check.argtypes = [ctypes.c_void_p]
    assert lib.mgn_debug_pack_check(eng.h) == 0


def test_set_params_with_unchanged_values_keeps_everything():
    """a caller that cannot tell whether its parameters changed sets them before every call: the same values are recognised (nothing is
    invalidated), new values are picked up by the next call of either kernel family"""
    cfg = cfg_dict(mps=2)
    ps = make_params(cfg, jitter=0.05)
    pos, s, r = small_mesh()
    N = pos.shape[0]
    nf, ef = random_inputs(N, s.size, cfg)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    o1 = eng.forward(nf, ef)
    eng.set_params(ps.copy())
    o2 = eng.forward(nf, ef)
    assert np.array_equal(o1, o2)
    ps2 = (ps * np.float32(1.01)).astype(np.float32)
    eng.set_params(ps2)
    o3 = eng.forward(nf, ef)
    ref = orc.forward(ps2, cfg, nf, ef, s, r)
    assert rel_max(o3, ref) <= TOL_15 and not np.array_equal(o3, o1)
    tgt = np.zeros((N, cfg["O"]), np.float32)
    mask = np.arange(N, dtype=np.int32)
    g1, l1 = eng.step(nf, ef, tgt, mask)
    eng.set_params(ps)                                   # back to the first values: the training kernels see them too
    g2, l2 = eng.step(nf, ef, tgt, mask)
    assert l1 != l2
    eng.set_params(ps.copy())
    g3, l3 = eng.step(nf, ef, tgt, mask)
    assert l3 == l2 and np.array_equal(g3, g2)

