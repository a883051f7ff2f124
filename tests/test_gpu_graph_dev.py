"""Graph prologue on the device (reference create_base_graph, src/graph.jl:25-55): integer results are compared
BIT FOR BIT with the oracle / the host versions; features within fp32 rounding."""
import time

import numpy as np
import pytest
import torch   # noqa: F401

import mgn_amd
import mgn_oracle as orc
from mgn_amd import synth
from util import cfg_dict, engine_for, make_params, rel_max

pytestmark = pytest.mark.gpu


def test_triangles_to_edges_dev_bit_exact():
    eng = engine_for(cfg_dict(mps=1))
    # KAT-1 shapes and a ragged soup (shared edges, repeated triangles, both orientations)
    for cells in (np.array([[0, 1, 2]], np.int32), np.array([[0, 1, 2], [1, 3, 2]], np.int32),
                  np.random.default_rng(0).integers(0, 40, (500, 3)).astype(np.int32)):
        s, r = eng.triangles_to_edges_dev(cells)
        so, ro = orc.triangles_to_edges(cells)
        assert np.array_equal(s, so) and np.array_equal(r, ro)
    # the M-1M cells (BASELINE.json configs[3]: 1 996 002 triangles -> 5 992 002 directed edges): device == host, bit for bit
    pos, cells = synth.grid_mesh(1000, 1000, 1234)
    t0 = time.perf_counter()
    sh, rh = mgn_amd.triangles_to_edges_native(cells)
    t_host = time.perf_counter() - t0
    eng.triangles_to_edges_dev(cells[:1000])          # warm-up (first-use allocations)
    t0 = time.perf_counter()
    sd, rd = eng.triangles_to_edges_dev(cells)
    t_dev = time.perf_counter() - t0
    assert sd.size == 5992002 and np.array_equal(sd, sh) and np.array_equal(rd, rh)
    print(f"triangles_to_edges on M-1M: host {t_host:.3f} s, device {t_dev:.3f} s (PCIe in / out included)")
    # too small a buffer: an error with the needed size, not an overrun
    n = mgn_amd._capi.C.c_int64()
    small = np.zeros(8, np.int32)
    rc = eng.lib.mgn_triangles_to_edges_dev(eng.h, cells.ctypes.data, 100, small.ctypes.data, small.ctypes.data, 4, mgn_amd._capi.C.byref(n))
    assert rc == -1 and n.value > 4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_set_static_mesh_equals_host_built_static_inputs(dtype):
    """one-hot + edge features computed on the device == create_base_graph's host arrays handed to mgn_set_static."""
    cfg = cfg_dict(mps=3)
    pos, cells, ntype, vel = synth.mesh_cyl(5, 700)
    s, r = synth.cells_to_edges(cells)
    N = pos.shape[0]
    ps = make_params(cfg)
    onehot = orc.one_hot(ntype, 7, 0).astype(np.float32)
    ef = orc.edge_features(pos, s, r).astype(np.float32)
    vm = np.isin(ntype, [0, 5]).astype(np.float32)
    norms = dict(node=(np.full(9, 0.7, np.float32), np.full(9, 0.1, np.float32)),
                 edge=(1.0 / np.maximum(ef.std(0), 1e-8), -ef.mean(0) / np.maximum(ef.std(0), 1e-8)), out=(np.full(2, 0.3, np.float32), np.zeros(2, np.float32)))
    x = vel.astype(np.float32)
    outs = []
    for dev in (False, True):
        eng = engine_for(cfg, dtype=dtype)
        eng.set_params(ps)
        eng.set_graph(s, r, N)
        eng.set_norms(**norms)
        if dev:
            eng.set_static_mesh(ntype, 0, 6, pos, vm)
        else:
            eng.set_static(onehot, ef, vm)
        outs.append([eng.ode_step(x) for _ in range(3)])
        eng.close()
    assert all(np.array_equal(a, outs[1][0]) for a in outs[1])
    # (the host arrays come from the float64 oracle rounded to fp32, the device computes in fp32: features differ by an ulp)
    assert rel_max(outs[1][0], outs[0][0]) <= (1e-5 if dtype == "f32" else 2e-2)


def test_world_edges_dev_equals_host_search_and_feeds_the_processor():
    m = synth.mesh_flag(3, 30, 24, radius=0.06)
    N = m["mesh_pos"].shape[0]
    cfg = dict(Fn=12, Fe=7, O=3, L=128, hidden_layers=2, mps=3)
    ps = orc.init_params(12, 7, 3, 128, 2, 3, 5, 0.1, Fe2=4)
    rng = np.random.default_rng(4)
    v0 = rng.standard_normal((N, 128)).astype(np.float32)
    e0 = rng.standard_normal((m["s"].size, 128)).astype(np.float32)
    nf = rng.standard_normal((N, 12)).astype(np.float32)
    for pos in (m["world_pos"], m["world_pos"][:, :2].copy()):
        for rad in (0.03, 0.06, 0.2):
            sh, rh = mgn_amd.world_edges_native(pos, rad, m["s"], m["r"])
            eng = mgn_amd.Engine(12, 7, 3, 128, 2, 3, Fe2=pos.shape[1] + 1)
            eng.set_graph(m["s"], m["r"], N)
            assert eng.world_edges_dev(1, pos, rad) == sh.size
            sd, rd = eng.edge_set_export(1)
            assert np.array_equal(sd, sh) and np.array_equal(rd, rh)
            eng.close()
    # the installed set drives the processor exactly like the host-installed one (same edge order -> same bits)
    pos, rad = m["world_pos"], 0.06
    sh, rh = mgn_amd.world_edges_native(pos, rad, m["s"], m["r"])
    rel = pos[sh] - pos[rh]
    ef2 = np.concatenate([rel, np.linalg.norm(rel, axis=1, keepdims=True)], 1).astype(np.float32)
    w0 = rng.standard_normal((sh.size, 128)).astype(np.float32)
    res = []
    for dev in (False, True):
        eng = mgn_amd.Engine(12, 7, 3, 128, 2, 3, Fe2=4)
        eng.set_params(ps)
        eng.set_graph(m["s"], m["r"], N)
        if dev:
            eng.world_edges_dev(1, pos, rad)
        else:
            eng.set_edge_set(1, sh, rh)
            eng.set_edge_features(1, ef2)
        out = eng.forward(nf, m["ef"])                     # world-edge features: device-computed vs host-computed
        eng.latents_import(v0, e0)
        eng.edge_latents_import(1, w0)
        eng.processor_steps_dev(3)
        v1, e1 = eng.latents_export()
        res.append((out, v1, e1, eng.edge_latents_export(1)))
        eng.close()
    assert rel_max(res[1][0], res[0][0]) <= 1e-5
    for a, b in zip(res[0][1:], res[1][1:]):
        assert np.array_equal(a, b)
    # scale + invariants: 200 k random points, no mesh edges
    big = np.random.default_rng(0).random((200000, 3)).astype(np.float32)
    eng = mgn_amd.Engine(12, 7, 3, 128, 2, 3, Fe2=4)
    eng.set_graph(np.zeros(0, np.int32), np.zeros(0, np.int32), 200000)
    t0 = time.perf_counter()
    nE = eng.world_edges_dev(1, big, 0.012)
    t_dev = time.perf_counter() - t0
    s3, r3 = eng.edge_set_export(1)
    t0 = time.perf_counter()
    sh, rh = mgn_amd.world_edges_native(big, 0.012, np.zeros(0, np.int32), np.zeros(0, np.int32))
    t_host = time.perf_counter() - t0
    assert nE == sh.size and np.array_equal(s3, sh) and np.array_equal(r3, rh)
    print(f"world edges, 200 k nodes: host search {t_host:.3f} s, device search + install {t_dev:.3f} s")
    with pytest.raises(mgn_amd.MgnError):
        eng.world_edges_dev(1, np.full((200000, 3), np.nan, np.float32), 0.1)
