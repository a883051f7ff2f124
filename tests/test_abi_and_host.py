"""CPU tests: the C-ABI library loads and exports every symbol include/mgn_hip.h declares; host logic
(parameter layout, receiver sort, CSR, RCB partition, halo lists) through host-only handles.
No compute call is made: there is no GPU here and the engine has no CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mgn_oracle as orc

import mgn_amd
from mgn_amd import MGN_DEVICE_NONE, Engine, MgnError, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "mgn_hip.h")).read()
    return set(re.findall(r"\b(mgn_[a-z0-9_]+)\s*\(", hdr))


def test_library_exports_every_declared_symbol(lib_built):
    lib = C.CDLL(lib_built)
    declared = header_symbols()
    assert len(declared) >= 38
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mgn_hip.h but not exported"
    assert declared == set(mgn_amd.PROTOTYPES), "ctypes prototypes and header disagree"


def test_param_count_and_config_validation(lib_built):
    lib = mgn_amd.load()
    for (Fn, Fe, O, L, mps) in [(9, 3, 2, 128, 15), (12, 7, 3, 64, 4), (9, 3, 2, 32, 1)]:
        cfg = mgn_amd._capi.MgnConfig(Fn, Fe, O, L, 2, mps, 0, 0, 1, -1)
        assert lib.mgn_param_count(C.byref(cfg)) == orc.param_count(Fn, Fe, O, L, 2, mps)
        cfg2 = mgn_amd._capi.MgnConfig(Fn, Fe, O, L, 2, mps, 0, 0, 1, -1, 2, 4)      # + world-edge set
        assert lib.mgn_param_count(C.byref(cfg2)) == orc.param_count(Fn, Fe, O, L, 2, mps, Fe2=4)
    for hl in (1, 3, 4):                  # reference Args.hidden_layers is any integer (src/MeshGraphNets.jl:35-38): 1 .. 4 here
        cfg = mgn_amd._capi.MgnConfig(9, 3, 2, 64, hl, 3, 0, 0, 1, -1)
        assert lib.mgn_param_count(C.byref(cfg)) == orc.param_count(9, 3, 2, 64, hl, 3)
    for bad in [dict(L=100), dict(hidden_layers=0), dict(hidden_layers=5), dict(mps=0), dict(Fn=0), dict(rank=2, nranks=2)]:
        kw = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=2, mps=2, rank=0, nranks=1, device=MGN_DEVICE_NONE)
        kw.update(bad)
        with pytest.raises(MgnError) as ei:
            Engine(**kw)
        assert ei.value.code == -1        # MGN_E_ARG


def test_no_gpu_means_no_compute(lib_built):
    """The product path must fail loudly without a device: no CPU fallback anywhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(MgnError) as ei:
        Engine(9, 3, 2)
    assert ei.value.code == -2            # MGN_E_HIP
    e = Engine(9, 3, 2, device=MGN_DEVICE_NONE)
    s, r = synth.random_graph(10, 30, 0)
    e.set_graph(s, r, 10)
    for call in (lambda: e.set_params(np.zeros(e.param_count, np.float32)),
                 lambda: e.forward(np.zeros((10, 9), np.float32), np.zeros((30, 3), np.float32)),
                 lambda: e.processor_steps_dev(1), lambda: e.proc_edge(0), lambda: e.latents_randn(1)):
        with pytest.raises(MgnError) as ei:
            call()
        assert ei.value.code == -2


def test_locality_restoring_node_order(lib_built):
    """mgn_set_graph numbers a mesh that arrives with scattered node labels in breadth-first order (graph_host.h), so that the ends of
    an edge are close in the engine's row order again -- what the sender gathers of the processor kernels need -- and leaves a
    coherently numbered mesh alone.  Invisible at the boundary: owned_nodes() is the map back."""
    pos, cells = synth.grid_mesh(60, 45, 1)
    s, r = synth.cells_to_edges(cells)
    N = pos.shape[0]

    def cost(e):
        own = e.owned_nodes()
        assert sorted(own.tolist()) == list(range(N))              # a permutation of all nodes
        g2l = np.empty(N, np.int64)
        g2l[own] = np.arange(N)
        snd, rcv, rowptr = e.local_graph()
        assert np.all(np.diff(rcv) >= 0) and np.array_equal(np.bincount(rcv, minlength=N), np.diff(rowptr))
        return float(np.abs(snd.astype(np.int64) - rcv).mean()), own, g2l

    e0 = Engine(9, 3, 2, device=MGN_DEVICE_NONE)
    e0.set_graph(s, r, N)
    c0, own0, _ = cost(e0)
    assert not _renumbered(e0) and np.array_equal(own0, np.arange(N))
    perm = np.random.default_rng(1).permutation(N).astype(np.int32)
    sp, rp = perm[s], perm[r]
    e1 = Engine(9, 3, 2, device=MGN_DEVICE_NONE)
    e1.set_graph(sp, rp, N)
    c1, own1, g2l1 = cost(e1)
    assert _renumbered(e1) and c1 <= 3 * c0                        # as local as the generator's own numbering (within a small factor)
    eid = e1.local_edges()
    snd, rcv, _ = e1.local_graph()
    assert np.array_equal(own1[snd], sp[eid]) and np.array_equal(own1[rcv], rp[eid])
    old = _renumber_mode(0)                                         # policy 0: never
    try:
        e2 = Engine(9, 3, 2, device=MGN_DEVICE_NONE)
        e2.set_graph(sp, rp, N)
        c2, own2, _ = cost(e2)
        assert not _renumbered(e2) and np.array_equal(own2, np.arange(N)) and c2 > 5 * c1
        _renumber_mode(2)                                           # policy 2: always
        e3 = Engine(9, 3, 2, device=MGN_DEVICE_NONE)
        e3.set_graph(s, r, N)
        assert _renumbered(e3)
    finally:
        _renumber_mode(old)
    # disconnected pieces and isolated nodes keep every node exactly once
    s4 = np.concatenate([sp, np.array([], np.int32)])
    e4 = Engine(9, 3, 2, device=MGN_DEVICE_NONE)
    e4.set_graph(s4, rp, N + 7)                                     # seven nodes without an edge
    assert sorted(e4.owned_nodes().tolist()) == list(range(N + 7))


def test_set_graph_argument_errors(lib_built):
    e = Engine(9, 3, 2, device=MGN_DEVICE_NONE)
    with pytest.raises(MgnError):
        e.set_graph(np.array([0, 9], np.int32), np.array([1, 1], np.int32), 3)            # out of range
    with pytest.raises(MgnError):
        e.set_graph(np.array([0], np.int32), np.array([1], np.int32), 3, index_base=2)
    with pytest.raises(ValueError):
        e.set_graph(np.array([0, 1], np.int32), np.array([1], np.int32), 3)
    e.set_graph(np.array([1, 3], np.int32), np.array([3, 1], np.int32), 3, index_base=1)  # Julia 1-based ok
    assert e.e_local == 2 and e.n_own == 3
    e.set_graph(np.zeros(0, np.int32), np.zeros(0, np.int32), 4)                           # empty edge set
    assert e.e_local == 0 and e.n_own == 4


def _renumber_mode(mode):
    lib = mgn_amd.load()
    lib.mgn_debug_renumber.restype = C.c_int
    lib.mgn_debug_renumber.argtypes = [C.c_int]
    return lib.mgn_debug_renumber(mode)


def _renumbered(e):
    e.lib.mgn_debug_renumbered.restype = C.c_int
    e.lib.mgn_debug_renumbered.argtypes = [C.c_void_p]
    return bool(e.lib.mgn_debug_renumbered(e.h))


@pytest.mark.parametrize("P", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("scattered", [False, True])
def test_partition_invariants(lib_built, P, scattered):
    pos, cells = synth.grid_mesh(23, 17, 5)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    if scattered:      # the same mesh under arbitrary node labels (DeepMind's trajectories; create_base_graph passes them through,
        perm = np.random.default_rng(3).permutation(N).astype(np.int32)     # reference src/graph.jl:30-36): label of old node i = perm[i]
        s, r = perm[s], perm[r]
        pos = pos[np.argsort(perm)]
    engs = []
    for rk in range(P):
        e = Engine(9, 3, 2, rank=rk, nranks=P, device=MGN_DEVICE_NONE)
        e.set_graph(s, r, N, mesh_pos=pos)
        engs.append(e)
    owner = engs[0].node_owner()
    counts = np.bincount(owner, minlength=P)
    assert counts.max() - counts.min() <= 1                       # RCB balances node counts
    seen_nodes, seen_edges = np.zeros(N, int), np.zeros(E, int)
    for rk, e in enumerate(engs):
        assert np.array_equal(e.node_owner(), owner)               # every rank computes the same partition
        own = e.owned_nodes()
        nb = e.boundary_count()
        assert np.all(owner[own] == rk)
        assert not (_renumbered(e) and not scattered)              # a coherent numbering is kept (a scattered one is replaced when
        if not _renumbered(e):                                     # that buys a factor RENUMBER_GAIN: not on a 50-node partition)
            assert np.all(np.diff(own[:nb]) > 0) and np.all(np.diff(own[nb:]) > 0)
        sidx_all = e.halo_send_index()
        assert set(sidx_all.tolist()) == set(range(nb))            # boundary nodes == send-listed nodes, numbered first
        seen_nodes[own] += 1
        eid = e.local_edges()
        seen_edges[eid] += 1
        snd, rcv, rowptr = e.local_graph()
        assert np.all(np.diff(rcv) >= 0)                           # receiver-sorted
        assert rowptr[0] == 0 and rowptr[-1] == eid.size
        assert np.array_equal(np.bincount(rcv, minlength=e.n_own), np.diff(rowptr))
        halo = e.halo_nodes()
        l2g = np.concatenate([own, halo])
        assert np.array_equal(l2g[snd], s[eid]) and np.array_equal(own[rcv], r[eid])   # local ids map back
        assert np.all(owner[halo] != rk)
        # stable: edges of one receiver keep their input order
        for n in range(min(e.n_own, 50)):
            seg = eid[rowptr[n]:rowptr[n + 1]]
            assert np.all(np.diff(seg) > 0)
        send, recv = e.halo_counts()
        assert send[rk] == 0 and recv[rk] == 0 and recv.sum() == e.n_halo
        assert np.array_equal(np.bincount(owner[halo], minlength=P), recv)
    assert np.all(seen_nodes == 1) and np.all(seen_edges == 1)
    for p in range(P):                                             # what p sends to q is what q expects from p
        sp, _ = engs[p].halo_counts()
        off = np.concatenate([[0], np.cumsum(sp)])
        sidx = engs[p].halo_send_index()
        own_p = engs[p].owned_nodes()
        for q in range(P):
            _, rq = engs[q].halo_counts()
            assert sp[q] == rq[p]
            roff = np.concatenate([[0], np.cumsum(rq)])
            assert np.array_equal(own_p[sidx[off[q]:off[q + 1]]], engs[q].halo_nodes()[roff[p]:roff[p + 1]])


@pytest.mark.parametrize("P", [2, 3, 8])
@pytest.mark.parametrize("scattered", [False, True])
def test_rank_local_ingest_builds_the_same_partition(lib_built, P, scattered):
    """mgn_partition_nodes + mgn_set_graph_local: a rank that is handed only the edges it has an end of (with their positions in the global
    list) ends up with the local graph mgn_set_graph derives from the global lists -- owned nodes and their order, halo rows, local edges
    and their global positions, CSR, send lists."""
    pos, cells = synth.grid_mesh(23, 17, 5)
    s, r = synth.cells_to_edges(cells)
    N = pos.shape[0]
    if scattered:
        perm = np.random.default_rng(3).permutation(N).astype(np.int32)
        s, r = perm[s], perm[r]
        pos = pos[np.argsort(perm)]
    owner = Engine.partition_nodes(N, P, mesh_pos=pos)
    for base in (0, 1):
        for rk in range(P):
            a = Engine(9, 3, 2, rank=rk, nranks=P, device=MGN_DEVICE_NONE)
            a.set_graph(s + base, r + base, N, index_base=base, mesh_pos=pos)
            assert np.array_equal(a.node_owner(), owner)
            b = Engine(9, 3, 2, rank=rk, nranks=P, device=MGN_DEVICE_NONE)
            b.set_graph_local(s + base, r + base, N, owner, index_base=base)
            assert (a.n_own, a.n_halo, a.e_local, a.boundary_count()) == (b.n_own, b.n_halo, b.e_local, b.boundary_count())
            assert np.array_equal(a.owned_nodes(), b.owned_nodes()) and np.array_equal(a.halo_nodes(), b.halo_nodes())
            assert np.array_equal(a.local_edges(), b.local_edges())
            for x, y in zip(a.local_graph(), b.local_graph()):
                assert np.array_equal(x, y)
            for x, y in zip(a.halo_counts(), b.halo_counts()):
                assert np.array_equal(x, y)
            assert np.array_equal(a.halo_send_index(), b.halo_send_index())
            assert a.edge_set_info(0) == b.edge_set_info(0)                       # (global E, local E)
            assert a.edge_boundary_tiles(0) == b.edge_boundary_tiles(0)
    # without positions: contiguous index blocks, as mgn_set_graph falls back to
    assert np.array_equal(Engine.partition_nodes(10, 4), np.array([0, 0, 0, 1, 1, 2, 2, 2, 3, 3], np.int32))
    e = Engine(9, 3, 2, rank=0, nranks=P, device=MGN_DEVICE_NONE)
    with pytest.raises(MgnError):                                              # positions that are not ascending
        gid = np.array([3, 1], np.int64)
        e._chk(e.lib.mgn_set_graph_local(e.h, N, owner.ctypes.data_as(C.POINTER(C.c_int32)), s.size, 2,
                                         s[:2].ctypes.data_as(C.POINTER(C.c_int32)), r[:2].ctypes.data_as(C.POINTER(C.c_int32)),
                                         gid.ctypes.data_as(C.POINTER(C.c_int64)), 0))
    bad = owner.copy()
    bad[0] = P
    with pytest.raises(MgnError):                                              # an owner outside [0, nranks)
        e.set_graph_local(s, r, N, bad)


@pytest.mark.parametrize("P", [1, 3])
def test_two_edge_sets_partition_union_halo(lib_built, P):
    """Second edge set (world edges): same node partition and order as with the mesh set alone when no halo changes;
    halo / send lists are the union over both sets; the set can be replaced and emptied."""
    m = synth.mesh_flag(3, 16, 12, radius=0.11)
    N, s, r, s2, r2 = m["mesh_pos"].shape[0], m["s"], m["r"], m["s2"], m["r2"]
    assert s2.size > 50
    engs = []
    for rk in range(P):
        e = Engine(12, 7, 3, rank=rk, nranks=P, device=MGN_DEVICE_NONE, Fe2=4)
        e.set_graph(s, r, N, mesh_pos=m["mesh_pos"])
        assert e.edge_set_info(1) == (0, 0)
        engs.append(e)
    owner = engs[0].node_owner()
    own_before = [e.owned_nodes() for e in engs]
    for e in engs:
        e.set_edge_set(1, s2, r2)
    assert sum(e.e_local for e in engs) == s.size and sum(e.edge_set_info(1)[1] for e in engs) == s2.size
    for rk, e in enumerate(engs):
        assert np.array_equal(e.node_owner(), owner)                        # partition kept
        assert e.edge_set_info(1)[0] == s2.size
        assert e.edge_set_info(1)[1] == int(np.sum(owner[r2] == rk))        # edges live with their receiver's owner
        assert sorted(e.owned_nodes().tolist()) == sorted(own_before[rk].tolist())
        both_s, both_r = np.concatenate([s, s2]), np.concatenate([r, r2])
        want_halo = np.unique(both_s[(owner[both_r] == rk) & (owner[both_s] != rk)])
        assert np.array_equal(np.sort(e.halo_nodes()), want_halo)            # union over the sets
        if P == 1:
            assert np.array_equal(e.owned_nodes(), own_before[rk]) and e.n_halo == 0
    for p in range(P):
        sp, _ = engs[p].halo_counts()
        for q in range(P):
            assert sp[q] == engs[q].halo_counts()[1][p]
    for e in engs:                                                           # empty it again: back to the mesh-only halo
        e.set_edge_set(1, s2[:0], r2[:0])
        assert e.edge_set_info(1) == (0, 0)
    for rk, e in enumerate(engs):
        want = np.unique(s[(owner[r] == rk) & (owner[s] != rk)])
        assert np.array_equal(np.sort(e.halo_nodes()), want)
    one = Engine(9, 3, 2, device=MGN_DEVICE_NONE)
    one.set_graph(s, r, N)
    with pytest.raises(MgnError):
        one.set_edge_set(1, s2, r2)                                          # single-edge-set handle


def test_partition_without_positions_uses_index_blocks(lib_built):
    s, r = synth.random_graph(100, 400, 1, allow_isolated=False)
    e = Engine(9, 3, 2, rank=1, nranks=4, device=MGN_DEVICE_NONE)
    e.set_graph(s, r, 100)
    assert sorted(e.owned_nodes().tolist()) == list(range(25, 50))


def test_native_graph_prologue_helpers(lib_built):
    """N3: mgn_triangles_to_edges / mgn_edge_features (host C++, scalable) against the oracle's reference-order
    restatement, incl. KAT-1 and KAT-3."""
    s, r = mgn_amd.triangles_to_edges_native(np.array([[0, 1, 2]], np.int32))
    assert s.size == 6 and set(zip(s.tolist(), r.tolist())) == {(1, 0), (2, 1), (2, 0), (0, 1), (1, 2), (0, 2)}
    s, r = mgn_amd.triangles_to_edges_native(np.array([[0, 1, 2], [1, 3, 2]], np.int32))
    assert s.size == 10
    pos, cells = synth.grid_mesh(37, 23, 4)
    s, r = mgn_amd.triangles_to_edges_native(cells)
    so, ro = orc.triangles_to_edges(cells)
    assert np.array_equal(s, so) and np.array_equal(r, ro)              # same first-occurrence order
    s1, r1 = mgn_amd.triangles_to_edges_native(cells + 1)               # 1-based cells (Julia data)
    assert np.array_equal(s1, so + 1) and np.array_equal(r1, ro + 1)
    ef = mgn_amd.edge_features_native(pos, s, r)
    assert np.allclose(ef, orc.edge_features(pos, s, r), atol=1e-6)
    ef345 = mgn_amd.edge_features_native(np.array([[0, 0], [3, 0], [3, 4]], np.float32), [2, 3, 3], [1, 2, 1], index_base=1)
    assert sorted(ef345[:, 2].tolist()) == [3.0, 4.0, 5.0]
    with pytest.raises(ValueError):
        mgn_amd.triangles_to_edges_native(np.zeros((3, 2), np.int32))


def test_world_edges_native_matches_dense_search(lib_built):
    """mgn_world_edges (uniform-grid search) against the dense O(N^2) NumPy search on the folded cloth, 2-D and 3-D, both
    index bases; and at a size the dense search cannot do (200 k nodes) by its invariants."""
    m = synth.mesh_flag(3, 30, 24, radius=0.06)
    for pos in (m["world_pos"], m["world_pos"][:, :2].copy()):
        for rad in (0.03, 0.06, 0.2):
            s0, r0 = synth.world_edges(pos, rad, m["s"], m["r"])
            s1, r1 = mgn_amd.world_edges_native(pos, rad, m["s"], m["r"])
            k0 = np.sort(r0.astype(np.int64) * 100000 + s0)
            assert np.array_equal(r1.astype(np.int64) * 100000 + s1, k0)          # receiver-major, senders ascending
            s2, r2 = mgn_amd.world_edges_native(pos, rad, m["s"] + 1, m["r"] + 1, index_base=1)
            assert np.array_equal(s2, s1 + 1) and np.array_equal(r2, r1 + 1)
    rng = np.random.default_rng(0)
    big = rng.random((200000, 3)).astype(np.float32)
    s3, r3 = mgn_amd.world_edges_native(big, 0.012, np.zeros(0, np.int32), np.zeros(0, np.int32))
    d = np.linalg.norm(big[s3] - big[r3], axis=1)
    assert s3.size > 100000 and d.max() < 0.012 and np.all(s3 != r3)
    key = r3.astype(np.int64) * 200000 + s3
    assert np.all(np.diff(key) > 0)                                               # sorted, no duplicates
    rev = np.isin(s3.astype(np.int64) * 200000 + r3, key)
    assert rev.all()                                                              # symmetric


@pytest.mark.parametrize("P", [2, 4, 8])
def test_edge_tiles_without_halo_senders(lib_built, P):
    """Interior / boundary split of the edge step: every edge with a halo sender lies in the first
    `boundary` 32-edge tiles; for a two-way mesh that is a small share of the tiles."""
    pos, cells = synth.grid_mesh(60, 50, 7)
    s, r = synth.cells_to_edges(cells)
    N = pos.shape[0]
    for rk in range(P):
        e = Engine(9, 3, 2, rank=rk, nranks=P, device=MGN_DEVICE_NONE)
        e.set_graph(s, r, N, mesh_pos=pos)
        tb, nt = e.edge_boundary_tiles()
        snd, rcv, rowptr = e.local_graph()
        assert nt == (snd.size + 31) // 32 and 0 < tb <= nt
        assert np.all(snd[tb * 32:] < e.n_own)                        # no halo sender beyond the boundary tiles
        assert np.any(snd[max(0, (tb - 1) * 32): tb * 32] >= e.n_own)  # and the last boundary tile has one
        assert np.all(rcv[snd >= e.n_own] < e.boundary_count())       # such edges end at boundary nodes (numbered first)
        assert tb <= max(4, nt // 3)
    one = Engine(9, 3, 2, device=MGN_DEVICE_NONE)
    one.set_graph(s, r, N)
    assert one.edge_boundary_tiles() == (0, (s.size + 31) // 32)


def test_c_abi_from_plain_c(lib_built, tmp_path):
    """include/mgn_hip.h compiles as pedantic C99 and the library is driven from a C program (no Python, no C++)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "c_abi", "abi_check.c")
    exe = str(tmp_path / "abi_check")
    libdir = os.path.dirname(lib_built)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", src, "-o", exe, "-L", libdir, "-lmgn_hip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "abi_check OK" in out.stdout, out.stdout + out.stderr


def test_integration_doc_names_every_symbol(lib_built):
    """INTEGRATION.md maps every entry point of include/mgn_hip.h to the reference interface it replaces: no symbol may be
    missing from it (a slash list such as `mgn_fwd_upload/encode/decode` counts)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = []
    for sym in sorted(header_symbols()):
        if sym in doc:
            continue
        tail = sym.split("_")[-1]
        if re.search(r"mgn_[a-z0-9_/]*/" + re.escape(tail) + r"\b", doc) or (sym.rsplit("_", 1)[0] + "_*") in doc:
            continue
        missing.append(sym)
    assert not missing, missing
