"""Randomised parity sweep (seeded, reproducible): widths, depths and ragged graphs the fixed tests do not name --
forward, device-resident processor steps, fused RHS and the training step against the float64 oracle.
Run on the MI355X box with `-m gpu`."""
import os

import numpy as np
import pytest
import torch   # before the engine's first HIP call (the fp32 cross-check below runs on cuda)

import mgn_amd
import mgn_oracle as orc
from mgn_amd import synth
from util import TOL_15, rel_max, set_kernel_path

pytestmark = pytest.mark.gpu
SWEEP = int(os.environ.get("MGN_SWEEP", "12"))   # more seeds for an occasional soak: MGN_SWEEP=200 pytest ...


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.choice([32, 64, 128]))
    cfg = dict(Fn=int(rng.integers(1, 17)), Fe=int(rng.integers(1, 9)), O=int(rng.integers(1, 5)), L=L, hidden_layers=2,
               mps=int(rng.integers(1, 5)))
    cfg["Fn"] = max(cfg["Fn"], cfg["O"])
    N = int(rng.integers(1, 400))
    E = int(rng.integers(0, 3500))
    s, r = synth.random_graph(N, E, seed, allow_isolated=bool(rng.integers(2)))
    if rng.integers(3) == 0 and E > 0:          # a hub: one receiver with hundreds of incoming edges
        r[: E // 3] = int(rng.integers(N))
    ps = orc.init_params(cfg["Fn"], cfg["Fe"], cfg["O"], L, 2, cfg["mps"], seed=seed, ln_jitter=0.1)
    return rng, cfg, N, E, s, r, ps


@pytest.mark.parametrize("seed", range(SWEEP))
def test_random_forward_and_processor(seed):
    rng, cfg, N, E, s, r, ps = draw(seed)
    path = int(rng.integers(0, 4)) if cfg["L"] == 128 else 0
    old = set_kernel_path(path)
    try:
        base = int(rng.integers(2))
        eng = mgn_amd.Engine(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], 2, cfg["mps"])
        eng.set_params(ps)
        eng.set_graph(s + base, r + base, N, index_base=base)
        nf = rng.standard_normal((N, cfg["Fn"])).astype(np.float32)
        ef = rng.standard_normal((E, cfg["Fe"])).astype(np.float32)
        out = eng.forward(nf, ef)
        ref = orc.forward(ps, cfg, nf, ef, s, r)
        assert rel_max(out, ref) <= TOL_15, (cfg, N, E, path, rel_max(out, ref))
        v = rng.standard_normal((N, cfg["L"])).astype(np.float32)
        e = rng.standard_normal((E, cfg["L"])).astype(np.float32)
        k = int(rng.integers(1, cfg["mps"] + 1))
        eng.latents_import(v, e)
        eng.processor_steps_dev(k)
        v1, e1 = eng.latents_export()
        rv, re = orc.processor_steps(ps, cfg, v, e, s, r, k)
        assert rel_max(v1, rv) <= TOL_15 and (E == 0 or rel_max(e1, re) <= TOL_15), (cfg, N, E, path, k)
    finally:
        set_kernel_path(old)


@pytest.mark.parametrize("seed", range(max(6, SWEEP // 2)))
def test_random_training_step(seed):
    rng, cfg, N, E, s, r, ps = draw(100 + seed)
    eng = mgn_amd.Engine(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], 2, cfg["mps"])
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    nf = rng.standard_normal((N, cfg["Fn"])).astype(np.float32)
    ef = rng.standard_normal((E, cfg["Fe"])).astype(np.float32)
    target = rng.standard_normal((N, cfg["O"])).astype(np.float32)
    mask = rng.choice(N, max(1, N // 2), replace=False).astype(np.int32)
    gs, loss = eng.step(nf, ef, target, mask)
    ref, ref_loss = orc.step_grads(ps, cfg, nf, ef, s, r, target, mask)
    assert abs(loss - ref_loss) <= 1e-5 * max(abs(ref_loss), 1e-6), (cfg, N, E)
    # fp32 against float64: a ReLU pre-activation within fp32 rounding of zero, or a hub with a thousand incoming edges,
    # can move the whole gradient by a few 1e-3 -- in EVERY fp32 implementation alike (seen: engine, PyTorch fp32 on GPU
    # and on CPU all 3.4e-3 from float64 and 3e-7 from each other).  So: close to float64, or else indistinguishable from
    # the independent PyTorch fp32 restatement.
    err64 = np.linalg.norm(gs - ref) / np.linalg.norm(ref)
    if err64 > 2e-3:
        import torch_reference as tr
        g32, _ = tr.step(ps, cfg, nf, ef, s, r, target, mask, dtype=torch.float32, device="cuda")
        err32 = np.linalg.norm(gs - g32) / np.linalg.norm(ref)
        if err32 > 1e-4:
            # the engine alone took the other branch of a ReLU (its summation order differs from PyTorch's too): then the
            # deviation is confined to this exact input and a 1e-4 jitter of the features removes it (seen: 3.3e-3 on
            # one draw, 2e-7 on five jittered copies); a real defect would survive the jitter
            assert err64 <= 5e-2, (cfg, N, E, err64)
            for trial in (1, 2):
                nf_j = (nf * (1.0 + 1e-4 * np.random.default_rng(trial).standard_normal(nf.shape))).astype(np.float32)
                gs_j, _ = eng.step(nf_j, ef, target, mask)
                ref_j, _ = orc.step_grads(ps, cfg, nf_j, ef, s, r, target, mask)
                assert np.linalg.norm(gs_j - ref_j) <= 2e-3 * np.linalg.norm(ref_j), (cfg, N, E, trial, err64, err32)
        else:
            assert err64 <= 5e-2, (cfg, N, E, err64, err32)


@pytest.mark.parametrize("seed", range(max(6, SWEEP // 4)))
def test_random_bf16_band(seed):
    """bf16 mode (L = 128) on the same random family: relative L2 of the latents within the stated 3e-2 band after up to 4
    steps."""
    rng = np.random.default_rng(5000 + seed)
    cfg = dict(Fn=int(rng.integers(2, 13)), Fe=int(rng.integers(1, 8)), O=int(rng.integers(1, 4)), L=128, hidden_layers=2,
               mps=int(rng.integers(1, 5)))
    N = int(rng.integers(2, 500))
    E = int(rng.integers(1, 4000))
    s, r = synth.random_graph(N, E, seed)
    ps = orc.init_params(cfg["Fn"], cfg["Fe"], cfg["O"], 128, 2, cfg["mps"], seed=seed, ln_jitter=0.05)
    v = rng.standard_normal((N, 128)).astype(np.float32)
    e = rng.standard_normal((E, 128)).astype(np.float32)
    eng = mgn_amd.Engine(cfg["Fn"], cfg["Fe"], cfg["O"], 128, 2, cfg["mps"], dtype="bf16")
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    eng.latents_import(v, e)
    eng.processor_steps_dev(cfg["mps"])
    v1, e1 = eng.latents_export()
    rv, re = orc.processor_steps(ps, cfg, v, e, s, r, cfg["mps"])
    l2 = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    assert l2(v1, rv) <= 3e-2 and l2(e1, re) <= 3e-2, (cfg, N, E, l2(v1, rv), l2(e1, re))


@pytest.mark.parametrize("seed", range(max(8, SWEEP // 4)))
def test_random_partitions_loopback(seed):
    """Random graphs cut into 2..5 index-block partitions (no positions; some parts may have no halo, no edges, or fewer
    than 32 nodes), driven by the staged driver with the split edge step: the merged result equals the single-domain oracle."""
    from importlib import import_module
    halo = import_module("mgn_amd.halo")
    rng = np.random.default_rng(9000 + seed)
    L = int(rng.choice([32, 64, 128]))
    cfg = dict(Fn=9, Fe=3, O=2, L=L, hidden_layers=2, mps=int(rng.integers(1, 4)))
    N = int(rng.integers(6, 300))
    E = int(rng.integers(0, 2500))
    P = int(rng.integers(2, 6))
    s, r = synth.random_graph(N, E, seed)
    ps = orc.init_params(9, 3, 2, L, 2, cfg["mps"], seed=seed, ln_jitter=0.1)
    v0 = rng.standard_normal((N, L)).astype(np.float32)
    e0 = rng.standard_normal((E, L)).astype(np.float32)
    stream = torch.cuda.current_stream().cuda_stream
    engs = []
    for k in range(P):
        g = mgn_amd.Engine(9, 3, 2, L, 2, cfg["mps"], rank=k, nranks=P)
        g.set_stream(stream)
        g.set_params(ps)
        g.set_graph(s, r, N)
        g.latents_import(v0, e0)
        engs.append(g)
    mgn_amd.run_processor_staged(engs, halo.LoopbackExchange(engs, torch.device("cuda")), cfg["mps"])
    torch.cuda.synchronize()
    v, e = np.zeros((N, L), np.float32), np.zeros((E, L), np.float32)
    for g in engs:
        g.latents_export(v, e)
    rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, cfg["mps"])
    assert rel_max(v, rv) <= TOL_15 and (E == 0 or rel_max(e, re) <= TOL_15), (cfg, N, E, P)
