"""GPU parity tests for the bf16 mode (BASELINE.json configs[2] precision): bf16 storage + bf16 MFMA in the processor (small graphs:
bf16 storage + fp32 MFMA in the 16-row kernels), fp32 accumulate / LayerNorm / residual / aggregation.  Stated tolerance: relative L2 <= 3e-2 against
the float64 oracle after 15 steps (LayerNorm re-centres every step)."""
from importlib import import_module

import numpy as np
import pytest
import torch

import mgn_oracle as orc
from util import cfg_dict, engine_for, make_params, random_inputs, small_mesh

import mgn_amd
from mgn_amd import synth

pytestmark = pytest.mark.gpu
TOL_BF16 = 3e-2


def rel_l2(a, ref):
    return float(np.linalg.norm(np.asarray(a, np.float64) - ref) / np.linalg.norm(ref))


@pytest.fixture(autouse=True, params=[0, 1], ids=["auto", "bf16-mfma"])
def bf16_family(request):
    """Small graphs in bf16 mode run the 16-row kernels on the bf16 arrays (fp32 weights and arithmetic) when the kernel family
    is chosen automatically; kernel path 1 keeps the bf16-MFMA kernels that large meshes use.  Every test here covers both."""
    from util import set_kernel_path
    old = set_kernel_path(request.param)
    yield request.param
    set_kernel_path(old)


@pytest.mark.parametrize("nsteps", [1, 15])
def test_bf16_processor_steps(nsteps):
    cfg = cfg_dict(mps=15)
    pos, cells, _, _ = synth.mesh_cyl(1234, 500)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg, jitter=0.05)
    rng = np.random.default_rng(3)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    eng = engine_for(cfg, dtype="bf16")
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    v1, e1 = eng.processor_steps(v, e, nsteps)
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, nsteps)
    assert rel_l2(v1, rv) <= TOL_BF16 and rel_l2(e1, re) <= TOL_BF16, (rel_l2(v1, rv), rel_l2(e1, re))
    # and it is genuinely bf16: not bit-identical to the fp32 engine
    f32 = engine_for(cfg)
    f32.set_params(ps)
    f32.set_graph(s, r, N)
    v2, _ = f32.processor_steps(v, e, nsteps)
    assert rel_l2(v2, rv) < rel_l2(v1, rv)


def test_bf16_forward_and_ragged():
    cfg = cfg_dict(mps=3)
    ps = make_params(cfg)
    for (N, E, seed) in [(5, 1, 1), (40, 700, 3), (70, 2049, 5)]:
        s, r = synth.random_graph(N, E, seed)
        if E >= 700:
            r[: E // 2] = 3
        nf, ef = random_inputs(N, E, cfg, seed)
        eng = engine_for(cfg, dtype="bf16")
        eng.set_params(ps)
        eng.set_graph(s, r, N)
        out = eng.forward(nf, ef)
        ref = orc.forward(ps, cfg, nf, ef, s, r)
        assert rel_l2(out, ref) <= TOL_BF16, (N, E, rel_l2(out, ref))
        assert np.array_equal(out, eng.forward(nf, ef))      # deterministic


def test_bf16_partitioned_equals_single():
    halo = import_module("mgn_amd.halo")
    cfg = cfg_dict(mps=4)
    pos, cells = synth.grid_mesh(40, 33, 9)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    rng = np.random.default_rng(1)
    v0 = rng.standard_normal((N, 128)).astype(np.float32)
    e0 = rng.standard_normal((E, 128)).astype(np.float32)
    single = engine_for(cfg, dtype="bf16")
    single.set_params(ps)
    single.set_graph(s, r, N)
    v1, e1 = single.processor_steps(v0, e0, 4)
    stream = torch.cuda.current_stream().cuda_stream
    engs = []
    for k in range(4):
        e = engine_for(cfg, rank=k, nranks=4, dtype="bf16")
        e.set_stream(stream)
        e.set_params(ps)
        e.set_graph(s, r, N, mesh_pos=pos)
        e.latents_import(v0, e0)
        engs.append(e)
    mgn_amd.run_processor_staged(engs, halo.LoopbackExchange(engs, torch.device("cuda")), 4)
    torch.cuda.synchronize()
    v, e = np.zeros((N, 128), np.float32), np.zeros((E, 128), np.float32)
    for g in engs:
        g.latents_export(v, e)
    # the per-receiver sums are rounded to bf16 once per edge tile / carry row: partitions change the tiling
    assert rel_l2(v, v1.astype(np.float64)) <= 1e-2 and rel_l2(e, e1.astype(np.float64)) <= 1e-2
    rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, 4)
    assert rel_l2(v, rv) <= TOL_BF16 and rel_l2(e, re) <= TOL_BF16


def test_bf16_config_errors():
    from mgn_amd import MgnError
    with pytest.raises(MgnError):
        mgn_amd.Engine(9, 3, 2, 64, 2, 2, dtype="bf16")
