"""The split path (csrc/split.hip; MGN_FP32_SPLIT / mgn_debug_fp32_split = 1, the default): the L x L layers of large fp32 launches on the
16-bit matrix cores at fp32 accuracy.  Default (mgn_debug_split_f16 = 1): every fp32 operand as two fp16 pieces after a power-of-two
scaling (per weight chunk, per activation row), three piece products (k_edge_ring_h, k_node_split_h, k_project_split_h).  With
mgn_debug_split_f16 = 0: three exact bf16 pieces, six of the nine piece products (k_edge_ring, k_node_split, k_project_split).  Both must meet the SAME tolerances against the float64 oracle as the fp32-MFMA kernels (it is not a reduced-precision mode), on a
mesh large enough for the persistent kernels to be chosen, with ragged receiver runs, and with two edge sets."""
import numpy as np
import pytest
import torch  # noqa: F401

import mgn_oracle as orc
from mgn_amd import synth
from util import TOL_15, TOL_STEP, cfg_dict, engine_for, make_params, rel_max, set_c16_row_tiles, set_c16_split, set_fp32_split, set_kernel_path, set_split_f16

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[100, 1], ids=["ring_h", "ring"])
def split_on(request):
    """100: the default -- mode 1 on two fp16 pieces and three piece products (k_edge_ring_h, k_node_split_h, k_project_split_h); 1: the same
    mode on three bf16 pieces and six products (k_edge_ring, k_node_split, k_project_split; mgn_debug_split_f16(0))"""
    old = set_fp32_split(1)
    oldh = set_split_f16(1 if request.param == 100 else 0)
    yield 1
    set_split_f16(oldh)
    set_fp32_split(old)


def _mesh(nx=150):
    pos, s, r = synth.mesh_1m(1234, nx, nx)          # 22 500 nodes, 133 k edges: 4 160 edge tiles > 16 per CU -> persistent kernels
    return pos, s, r


def test_split_numerics_of_one_layer_on_the_host():
    """the arithmetic itself, in numpy: three-way bf16 split is exact, six terms beat a plain fp32 GEMM against float64"""
    rng = np.random.default_rng(0)

    def bf16(x):
        u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)

    def split3(x):
        a = bf16(x); r1 = (x - a).astype(np.float32); b = bf16(r1); r2 = (r1 - b).astype(np.float32)
        return a, b, bf16(r2)

    X = rng.standard_normal((2048, 128)).astype(np.float32)
    W = (rng.uniform(-1, 1, (128, 128)) * np.sqrt(6 / 256)).astype(np.float32)
    x1, x2, x3 = split3(X); w1, w2, w3 = split3(W)
    assert np.array_equal((x1.astype(np.float64) + x2 + x3).astype(np.float32), X)
    ref = X.astype(np.float64) @ W.astype(np.float64)
    acc = np.zeros_like(X)
    for a, b in ((x1, w3), (x2, w2), (x3, w1), (x1, w2), (x2, w1), (x1, w1)):
        acc = (acc + a @ b).astype(np.float32)
    e_split = np.abs(acc - ref).max() / np.abs(ref).max()
    e_f32 = np.abs((X @ W) - ref).max() / np.abs(ref).max()
    assert e_split <= 4e-7 and e_split <= 2 * e_f32, (e_split, e_f32)


def test_two_fp16_pieces_numerics_of_one_layer_on_the_host():
    """the arithmetic of the default split path, in numpy: an operand times a power of two that puts the row's (chunk's) largest entry
    into [2^14, 2^15) is hi + lo in fp16 to 2^-23; hi x hi + hi x lo + lo x hi beat a plain fp32 GEMM against float64 whatever the
    scale of the inputs -- per ROW, which is what the kernels do -- and never overflow"""
    rng = np.random.default_rng(0)

    def scale(amax):
        e = np.floor(np.log2(np.maximum(amax, 2.0 ** -40)))
        return (2.0 ** (14 - e)).astype(np.float32)

    def split2(x, s):
        xs = (x * s).astype(np.float32)
        hi = xs.astype(np.float16)
        lo = (xs - hi.astype(np.float32)).astype(np.float16)
        assert np.isfinite(hi.astype(np.float32)).all()
        return hi.astype(np.float64), lo.astype(np.float64)

    W = (rng.uniform(-1, 1, (128, 128)) * np.sqrt(6 / 256)).astype(np.float32)
    sw = scale(np.abs(W).max())
    wh, wl = split2(W, sw)
    for row_scales in (np.ones(2048), 10.0 ** rng.uniform(-6, 6, 2048)):
        X = (rng.standard_normal((2048, 128)) * row_scales[:, None]).astype(np.float32)
        X[:, 64:] = np.maximum(X[:, 64:], 0)                                   # half of it ReLU-like
        sx = scale(np.abs(X).max(1, keepdims=True))
        xh, xl = split2(X, sx)
        err_rep = np.abs((xh + xl) / sx - X).max(1) / np.abs(X).max(1)
        assert err_rep.max() <= 2.0 ** -23
        ref = X.astype(np.float64) @ W.astype(np.float64)
        got = ((xh @ wl + xl @ wh + xh @ wh).astype(np.float32) / (sx * sw)).astype(np.float32)
        den = np.abs(ref).max(1, keepdims=True)                                 # per row: the rows differ by twelve orders of magnitude
        e_h2 = (np.abs(got - ref) / den).max()
        e_f32 = (np.abs((X @ W) - ref) / den).max()
        assert e_h2 <= 2.5e-7 and e_h2 <= e_f32, (e_h2, e_f32)


@pytest.mark.parametrize("nsteps,tol", [(1, TOL_STEP), (15, TOL_15)])
def test_split_edge_kernel_meets_the_fp32_tolerances(split_on, nsteps, tol):
    cfg = cfg_dict(mps=15)
    pos, s, r = _mesh()
    N, E = pos.shape[0], s.size
    ps = make_params(cfg, jitter=0.05)
    rng = np.random.default_rng(5)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, nsteps)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    v1, e1 = eng.processor_steps(v, e, nsteps)
    err_split = (rel_max(v1, rv), rel_max(e1, re))
    assert max(err_split) <= tol, err_split
    # the device-resident entry (hipGraph replay) gives the same bits
    eng.latents_import(v, e)
    eng.processor_steps_dev(nsteps)
    v2, e2 = eng.latents_export()
    assert np.array_equal(v2, v1) and np.array_equal(e2, e1)
    # and it is as close to float64 as the fp32-MFMA kernel is (not a reduced-precision mode)
    set_fp32_split(0)
    f32 = engine_for(cfg)
    f32.set_params(ps)
    f32.set_graph(s, r, N)
    v3, e3 = f32.processor_steps(v, e, nsteps)
    set_fp32_split(split_on)
    err_f32 = (rel_max(v3, rv), rel_max(e3, re))
    assert max(err_split) <= 2.0 * max(err_f32) + 1e-7, (err_split, err_f32)
    assert not np.array_equal(e3, e1)          # (it really is the other kernel)


@pytest.mark.parametrize("nx", [72, 100, 180])
def test_split_ring_block_shapes(nx):
    """the ring kernel's launch shapes by size: 3 - 20 tiles per CU run four-wave blocks (one partly filled round at 72 x 72, three
    rounds at 100 x 100), above that eight-wave blocks (180 x 180: 6 040 tiles); below, the cooperative kernels keep the launch"""
    cfg = cfg_dict(mps=2)
    pos, s, r = synth.mesh_1m(77, nx, nx)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg, jitter=0.05)
    rng = np.random.default_rng(nx)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    v1, e1 = eng.processor_steps(v, e, 2)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 2)
    assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15, (rel_max(v1, rv), rel_max(e1, re))
    old = set_fp32_split(0)
    try:
        f32 = engine_for(cfg)
        f32.set_params(ps)
        f32.set_graph(s, r, N)
        v0, e0 = f32.processor_steps(v, e, 2)
    finally:
        set_fp32_split(old)
    assert not np.array_equal(e0, e1)              # (the split path did run at this size)


@pytest.mark.parametrize("nx", [128, 354])
def test_processor_pass_is_bitwise_repeatable(nx):
    """the same latents through the same 15 steps, five times: bit-identical every time.  The streamed kernels pass every weight piece
    through a three-buffer LDS ring and park the tile's rows in LDS; a window overwritten before its last reader would show up
    here as a difference between repeats (tools/soak_determinism.py: the same at five sizes up to M-1M, twelve repeats)"""
    cfg = cfg_dict(mps=15)
    pos, s, r = synth.mesh_1m(5, nx, nx)
    N = pos.shape[0]
    eng = engine_for(cfg)
    eng.set_params(make_params(cfg, jitter=0.05))
    eng.set_graph(s, r, N)
    ref = None
    for k in range(5):
        eng.latents_randn(11)
        eng.processor_steps_dev(15)
        v, e = eng.latents_export()
        if ref is None:
            ref = (v.copy(), e.copy())
            assert np.isfinite(v).all() and np.isfinite(e).all()
        else:
            assert np.array_equal(v, ref[0]) and np.array_equal(e, ref[1]), k


@pytest.mark.parametrize("E", [140001, 200003])
def test_split_ragged_receivers(split_on, E):
    """hub nodes (runs that straddle many tiles), empty receivers, a last partial tile; 140 001 edges run the ring kernel in four-wave
    blocks, 200 003 in eight-wave blocks"""
    cfg = cfg_dict(mps=3)
    N = 9000
    s, r = synth.random_graph(N, E, 7)
    r[: E // 3] = 17
    r[E // 3: E // 3 + 5000] = 4000
    ps = make_params(cfg)
    rng = np.random.default_rng(2)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    eng = engine_for(cfg)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    v1, e1 = eng.processor_steps(v, e, 3)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 3)
    assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15, (rel_max(v1, rv), rel_max(e1, re))


def test_split_two_edge_sets(split_on):
    """a cloth large enough for the persistent kernels (160 x 160: 25 600 nodes, mesh + world edges): the split edge kernel per edge set"""
    import mgn_amd
    m = synth.mesh_flag(nx=160, ny=160, radius=0.012)
    N, E, E2 = m["mesh_pos"].shape[0], m["s"].size, m["s2"].size
    assert E > 16 * 256 * 32 and E2 > 1000                      # the mesh set is beyond the cooperative kernels' range
    cfg = dict(Fn=12, Fe=7, O=3, L=128, hidden_layers=2, mps=2, Fe2=4)
    ps = orc.init_params(12, 7, 3, 128, 2, 2, 1234, 0.05, Fe2=4)
    rng = np.random.default_rng(11)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    e2 = rng.standard_normal((E2, 128)).astype(np.float32)
    eng = mgn_amd.Engine(12, 7, 3, 128, 2, 2, Fe2=4)
    eng.set_params(ps)
    eng.set_graph(m["s"], m["r"], N)
    eng.set_edge_set(1, m["s2"], m["r2"])
    eng.latents_import(v, e)
    eng.edge_latents_import(1, e2)
    eng.processor_steps_dev(2)
    v1, e1 = eng.latents_export()
    e21 = eng.edge_latents_export(1)
    rv, re, re2 = orc.processor_steps(ps, cfg, v, e, m["s"], m["r"], 2, set2=(e2, m["s2"], m["r2"]))
    assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15 and rel_max(e21, re2) <= TOL_15


def test_split_two_edge_sets_node_side():
    """78 400 nodes (2 450 node tiles, beyond the cooperative kernels): the node MLP with the second set's aggregate block and both
    projections run on the split path (k_node_split<true>: the fifth chunk's pieces all stream from L2)"""
    import ctypes
    import mgn_amd
    m = synth.mesh_flag(nx=280, ny=280, radius=0.007)
    N, E, E2 = m["mesh_pos"].shape[0], m["s"].size, m["s2"].size
    assert N > 2048 * 32 and E2 > 1000
    cfg = dict(Fn=12, Fe=7, O=3, L=128, hidden_layers=2, mps=2, Fe2=4)
    ps = orc.init_params(12, 7, 3, 128, 2, 2, 1234, 0.05, Fe2=4)
    rng = np.random.default_rng(12)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    e2 = rng.standard_normal((E2, 128)).astype(np.float32)
    eng = mgn_amd.Engine(12, 7, 3, 128, 2, 2, Fe2=4)
    eng.set_params(ps)
    eng.set_graph(m["s"], m["r"], N)
    eng.set_edge_set(1, m["s2"], m["r2"])
    eng.latents_import(v, e)
    eng.edge_latents_import(1, e2)
    eng.processor_steps_dev(2)
    lib = mgn_amd.load()
    lib.mgn_debug_last_node_kernel.restype = ctypes.c_int
    assert lib.mgn_debug_last_node_kernel() == 6          # (kernels.hip, launch_node_step: k_node_split with two sets)
    v1, e1 = eng.latents_export()
    e21 = eng.edge_latents_export(1)
    rv, re, re2 = orc.processor_steps(ps, cfg, v, e, m["s"], m["r"], 2, set2=(e2, m["s2"], m["r2"]))
    assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15 and rel_max(e21, re2) <= TOL_15
    old = set_fp32_split(0)                                # and as close to float64 as the fp32-MFMA kernels
    try:
        f32 = mgn_amd.Engine(12, 7, 3, 128, 2, 2, Fe2=4)
        f32.set_params(ps)
        f32.set_graph(m["s"], m["r"], N)
        f32.set_edge_set(1, m["s2"], m["r2"])
        f32.latents_import(v, e)
        f32.edge_latents_import(1, e2)
        f32.processor_steps_dev(2)
        v0, _ = f32.latents_export()
    finally:
        set_fp32_split(old)
    assert rel_max(v1, rv) <= 2.0 * rel_max(v0, rv) + 1e-7 and not np.array_equal(v0, v1)


@pytest.mark.parametrize("rt", [0, 1, 2, 3])
def test_split_16_row_kernels_on_the_cylinder_mesh(rt):
    """cylinder_flow-sized mesh (N = 2 000, E = 11 954): the 16-row cooperative kernels on the split path (v_mfma_f32_16x16x32_f16, two fp16
    pieces with a scale per (row, k-step) exchanged through LDS; mgn_debug_split_f16(0): v_mfma_f32_16x16x32_bf16, three pieces) against the float64 oracle at the fp32 tolerances, for every number of row tiles per block, and no worse than
    twice the error of the same kernels on the fp32 MFMA pipe"""
    import ctypes
    import mgn_amd
    pos, cells, ntype, vel = synth.mesh_cyl(1234, 2000)
    s, r = synth.cells_to_edges(cells)
    cfg = cfg_dict(mps=15)
    ps = make_params(cfg, jitter=0.05)
    rng = np.random.default_rng(7)
    v = rng.standard_normal((2000, 128)).astype(np.float32)
    e = rng.standard_normal((s.size, 128)).astype(np.float32)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 15)
    lib = mgn_amd.load()
    lib.mgn_debug_last_node_kernel.restype = ctypes.c_int
    lib.mgn_debug_last_edge_kernel.restype = ctypes.c_int
    old_p, old_rt = set_kernel_path(5 if rt else 0), set_c16_row_tiles(rt)
    try:
        out = {}
        for on in (7, 0):                                    # every 16-row kernel on the split path / none
            old = set_c16_split(on)
            try:
                eng = engine_for(cfg)
                eng.set_params(ps)
                eng.set_graph(s, r, 2000)
                out[on] = eng.processor_steps(v, e, 15)
                fam = (lib.mgn_debug_last_edge_kernel(), lib.mgn_debug_last_node_kernel())
                assert fam == ((15, 8) if on else (2, 2)), fam          # 15: the 16-row edge kernel on two fp16 pieces (12: on three bf16 pieces)
            finally:
                set_c16_split(old)
        err = {on: max(rel_max(out[on][0], rv), rel_max(out[on][1], re)) for on in out}
        assert err[7] <= TOL_15 and err[0] <= TOL_15, err
        assert err[7] <= 2.0 * err[0] + 1e-7, err
        assert not np.array_equal(out[7][1], out[0][1])
    finally:
        set_kernel_path(old_p)
        set_c16_row_tiles(old_rt)


@pytest.mark.parametrize("n_pts", [2300, 3200, 3900], ids=["four row tiles", "five", "six"])
def test_split_16_row_edge_kernel_with_four_to_six_row_tiles(n_pts):
    """12.3 k .. 24.5 k edges: the 16-row edge kernel keeps one round of blocks by taking four to six row tiles per block (split path only;
    the pieces then stay in the exchange buffer and the chains read one k-step at a time)"""
    import ctypes
    import mgn_amd
    pos, cells, _, _ = synth.mesh_cyl(77, n_pts)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    assert 256 * 3 * 16 < E <= 256 * 6 * 16
    cfg = cfg_dict(mps=4)
    ps = make_params(cfg, jitter=0.05)
    rng = np.random.default_rng(n_pts)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 4)
    lib = mgn_amd.load()
    lib.mgn_debug_last_edge_kernel.restype = ctypes.c_int
    lib.mgn_debug_c16_edge_tiles.restype = ctypes.c_int
    lib.mgn_debug_c16_edge_tiles.argtypes = [ctypes.c_int]
    # since round 6 a one-set fp32 handle leaves the 16-row kernels at two edge tiles per CU (k_edge_ring_hs takes over: its LDS prologue is
    # 28 KiB); the five- and six-row-tile blocks are what other handles (two edge sets, bf16 storage) run up to three tiles per CU
    if n_pts != 2300:
        eng = engine_for(cfg)
        eng.set_params(ps)
        eng.set_graph(s, r, N)
        vr, er_ = eng.processor_steps(v, e, 4)
        assert lib.mgn_debug_last_edge_kernel() == 17                                 # k_edge_ring_hs<4>
        assert rel_max(vr, rv) <= TOL_15 and rel_max(er_, re) <= TOL_15, (rel_max(vr, rv), rel_max(er_, re))
    old_lim = lib.mgn_debug_c16_edge_tiles(3)
    try:
        eng = engine_for(cfg)
        eng.set_params(ps)
        eng.set_graph(s, r, N)
        v1, e1 = eng.processor_steps(v, e, 4)
    finally:
        lib.mgn_debug_c16_edge_tiles(old_lim)
    assert lib.mgn_debug_last_edge_kernel() == (15 if n_pts == 2300 else 12)      # four row tiles on two fp16 pieces; five and six on three bf16 pieces (two spill)
    assert rel_max(v1, rv) <= TOL_15 and rel_max(e1, re) <= TOL_15, (rel_max(v1, rv), rel_max(e1, re))
    old = set_c16_split(0)
    old_lim = lib.mgn_debug_c16_edge_tiles(3)
    try:
        f32 = engine_for(cfg)
        f32.set_params(ps)
        f32.set_graph(s, r, N)
        v0, e0 = f32.processor_steps(v, e, 4)
    finally:
        set_c16_split(old)
        lib.mgn_debug_c16_edge_tiles(old_lim)
    assert max(rel_max(v1, rv), rel_max(e1, re)) <= 2.0 * max(rel_max(v0, rv), rel_max(e0, re)) + 1e-7
    assert not np.array_equal(e0, e1)


@pytest.mark.parametrize("mesh", ["persistent", "16-row"])
def test_two_fp16_pieces_hold_rows_and_chunks_of_any_scale(mesh):
    """What the power-of-two scaling of the fp16 split has to deliver: rows whose magnitudes are spread over eight orders, rows of zeros,
    a row with one huge outlier, and weight chunks scaled apart by 10^6 -- every ROW of the result as accurate, relative to ITS largest
    entry, as the fp32-MFMA kernels make it (a global max-norm would only see the largest rows), and nothing overflows."""
    cfg = cfg_dict(mps=2)
    if mesh == "persistent":
        pos, s, r = synth.mesh_1m(5, 150, 150)
        N = pos.shape[0]
    else:
        pos, cells, _, _ = synth.mesh_cyl(1234, 2000)
        s, r = synth.cells_to_edges(cells)
        N = pos.shape[0]
    E = s.size
    ps = make_params(cfg, jitter=0.05).copy()
    lay = orc.model_layout(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], cfg["hidden_layers"], cfg["mps"])
    off = 0
    rng = np.random.default_rng(11)
    for bname, tensors in lay:                                   # the chunks of the processor MLPs scaled apart (biases with their layer)
        for tname, shape in tensors:
            n = int(np.prod(shape))
            if bname.startswith("proc") and tname in ("W1", "W2", "b1", "b2"):
                ps[off:off + n] *= np.float32({"W1": 1e-3, "b1": 1e-3, "W2": 1e3, "b2": 1.0}[tname])
            off += n
    v = rng.standard_normal((N, 128)).astype(np.float32) * (10.0 ** rng.uniform(-4, 4, (N, 1))).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32) * (10.0 ** rng.uniform(-4, 4, (E, 1))).astype(np.float32)
    v[::97] = 0.0
    e[::89] = 0.0
    e[5::101, 17] = 3.0e7                                         # one outlier in an otherwise ordinary row
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, 2)

    def row_err(a, ref):
        den = np.maximum(np.abs(ref).max(1), 1e-30)
        return float((np.abs(a.astype(np.float64) - ref).max(1) / den).max())

    out = {}
    for name, (split, f16) in {"fp32_mfma": (0, 0), "f16x2": (1, 1), "bf16x3": (1, 0)}.items():
        old, oldh = set_fp32_split(split), set_split_f16(f16)
        olds = set_c16_split(3 if split else 0)
        try:
            eng = engine_for(cfg)
            eng.set_params(ps)
            eng.set_graph(s, r, N)
            v1, e1 = eng.processor_steps(v, e, 2)
            assert np.isfinite(v1).all() and np.isfinite(e1).all(), name
            out[name] = max(row_err(v1, rv), row_err(e1, re))
            eng.close()
        finally:
            set_c16_split(olds)
            set_split_f16(oldh)
            set_fp32_split(old)
    assert out["fp32_mfma"] <= 2e-5, out
    assert out["f16x2"] <= 2e-5 and out["f16x2"] <= 2.0 * out["fp32_mfma"] + 1e-6, out
    assert out["bf16x3"] <= 2e-5, out
