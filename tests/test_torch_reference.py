"""An independent PyTorch restatement (tests/torch_reference.py) against the NumPy oracle (CPU), and against the HIP engine
in fp32 on the GPU box: three implementations of MGN-spec v1 written separately have to agree."""
import numpy as np
import pytest
import torch

import mgn_oracle as orc
import torch_reference as tr
from mgn_amd import synth
from util import TOL_15, cfg_dict, make_params, rel_max, small_mesh


def _problem(cfg, seed=0):
    pos, s, r = small_mesh(9, 7)
    N, E = pos.shape[0], s.size
    rng = np.random.default_rng(seed)
    return (s, r, rng.standard_normal((N, cfg["Fn"])), rng.standard_normal((E, cfg["Fe"])), rng.standard_normal((N, cfg["O"])),
            np.sort(rng.choice(N, N // 2, replace=False)))


@pytest.mark.parametrize("L,mps", [(32, 2), (128, 3)])
def test_torch_forward_and_autograd_match_oracle(L, mps):
    cfg = cfg_dict(L=L, mps=mps)
    ps = make_params(cfg).astype(np.float64)
    s, r, nf, ef, target, mask = _problem(cfg)
    out_t = tr.forward(torch.tensor(ps), cfg, torch.tensor(nf), torch.tensor(ef), s, r).numpy()
    assert rel_max(out_t, orc.forward(ps, cfg, nf, ef, s, r)) < 1e-12
    gs_t, loss_t = tr.step(ps, cfg, nf, ef, s, r, target, mask)
    gs_o, loss_o = orc.step_grads(ps, cfg, nf, ef, s, r, target, mask)
    assert abs(loss_t - loss_o) < 1e-12 * abs(loss_o)
    assert np.abs(gs_t - gs_o).max() <= 1e-10 * np.abs(gs_o).max()      # autograd == the hand-written reverse mode


def test_torch_two_edge_sets_match_oracle():
    m = synth.mesh_flag(4, 10, 8, radius=0.15)
    cfg = dict(Fn=12, Fe=7, O=3, L=32, hidden_layers=2, mps=2, Fe2=4)
    ps = orc.init_params(12, 7, 3, 32, 2, 2, seed=3, ln_jitter=0.1, Fe2=4).astype(np.float64)
    rng = np.random.default_rng(1)
    N = m["mesh_pos"].shape[0]
    nf = rng.standard_normal((N, 12))
    ef, ef2 = m["ef"].astype(np.float64), m["ef2"].astype(np.float64)
    out_t = tr.forward(torch.tensor(ps), cfg, torch.tensor(nf), torch.tensor(ef), m["s"], m["r"],
                       set2=(torch.tensor(ef2), m["s2"], m["r2"])).numpy()
    out_o = orc.forward(ps, cfg, nf, ef, m["s"], m["r"], set2=(ef2, m["s2"], m["r2"]))
    assert rel_max(out_t, out_o) < 1e-12


@pytest.mark.gpu
def test_hip_engine_matches_torch_fp32_on_gpu():
    """The 'plain PyTorch fp32 reference of the same op', run on the same GPU: forward and training step."""
    import mgn_amd
    cfg = cfg_dict(mps=15)
    pos, cells, node_type, vel = synth.mesh_cyl(1234, 800)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg, jitter=0.05)
    rng = np.random.default_rng(5)
    nf = rng.standard_normal((N, 9)).astype(np.float32)
    ef = rng.standard_normal((E, 3)).astype(np.float32)
    target = rng.standard_normal((N, 2)).astype(np.float32)
    mask = np.nonzero(np.isin(node_type, [0, 5]))[0].astype(np.int32)
    dev = torch.device("cuda")
    with torch.no_grad():
        out_t = tr.forward(torch.tensor(ps, device=dev), cfg, torch.tensor(nf, device=dev), torch.tensor(ef, device=dev), s, r).cpu().numpy()
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 15)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    out = eng.forward(nf, ef)
    ref64 = orc.forward(ps, cfg, nf, ef, s, r)
    assert rel_max(out, ref64) <= TOL_15
    assert rel_max(out, out_t) <= 2 * TOL_15                     # two fp32 implementations: both errors add
    gs, loss = eng.step(nf, ef, target, mask)
    gs_t, loss_t = tr.step(ps, cfg, nf, ef, s, r, target, mask, dtype=torch.float32, device="cuda")
    assert abs(loss - loss_t) <= 1e-4 * abs(loss_t)
    assert np.linalg.norm(gs - gs_t) <= 5e-3 * np.linalg.norm(gs_t)
