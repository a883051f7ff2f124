"""CPU test of the N > 1 path: world_size 2 over gloo.  Each rank builds ITS partition with the C library
(host-only handle), runs the staged driver mgn_amd.run_processor_staged with halo rows exchanged by
DistExchange (torch.distributed all_to_all_single), compute served by the NumPy stand-in engine.  The merged
result must equal the single-domain float64 oracle.  The same driver and exchange code runs on RCCL."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, tmp, nsteps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import mgn_amd
    import mgn_oracle as orc
    from importlib import import_module
    from rank_engine import OracleRankEngine
    halo = import_module("mgn_amd.halo")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = dict(Fn=9, Fe=3, O=2, L=32, hidden_layers=2, mps=nsteps)
    pos, cells = mgn_amd.synth.grid_mesh(13, 9, 2)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = orc.init_params(9, 3, 2, 32, 2, nsteps, 11, 0.1)
    rng = np.random.default_rng(3)
    v0, e0 = rng.standard_normal((N, 32)), rng.standard_normal((E, 32))
    eng = OracleRankEngine(cfg, ps, s, r, N, pos, rank, world)
    eng.latents_import(v0, e0)
    ex = halo.DistExchange(eng, torch.device("cpu"))
    mgn_amd.run_processor_staged([eng], ex, nsteps)
    v, e = np.zeros((N, 32)), np.zeros((E, 32))
    eng.latents_export(v, e)
    tv, te = torch.from_numpy(v), torch.from_numpy(e)
    dist.all_reduce(tv)     # owned rows are disjoint: the sum merges the partitions
    dist.all_reduce(te)
    if rank == 0:
        rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, nsteps)
        np.savez(os.path.join(tmp, "res.npz"), dv=np.abs(tv.numpy() - rv).max() / np.abs(rv).max(),
                 de=np.abs(te.numpy() - re).max() / np.abs(re).max(), n_halo=eng.n_halo)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_staged_processor_over_gloo(world):
    port = 29500 + (os.getpid() % 1000) + world
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, port, tmp, 3), nprocs=world, join=True)
        res = np.load(os.path.join(tmp, "res.npz"))
        assert res["n_halo"] > 0
        # halo rows travel as float32 (the wire format of the real engine): ~1e-8 relative, not 1e-16
        assert res["dv"] < 1e-6 and res["de"] < 1e-6, (res["dv"], res["de"])


def test_loopback_exchange_numpy_engines():
    """Same check in one process with LoopbackExchange (the single-GPU KAT-7 harness), 4 partitions."""
    sys.path.insert(0, HERE)
    import mgn_amd
    import mgn_oracle as orc
    from importlib import import_module
    from rank_engine import OracleRankEngine
    halo = import_module("mgn_amd.halo")
    cfg = dict(Fn=9, Fe=3, O=2, L=32, hidden_layers=2, mps=2)
    pos, cells = mgn_amd.synth.grid_mesh(11, 10, 4)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = orc.init_params(9, 3, 2, 32, 2, 2, 5, 0.1)
    rng = np.random.default_rng(8)
    v0, e0 = rng.standard_normal((N, 32)), rng.standard_normal((E, 32))
    engs = [OracleRankEngine(cfg, ps, s, r, N, pos, k, 4) for k in range(4)]
    for e in engs:
        e.latents_import(v0, e0)
    mgn_amd.run_processor_staged(engs, halo.LoopbackExchange(engs, torch.device("cpu")), 2)
    v, e = np.zeros((N, 32)), np.zeros((E, 32))
    for g in engs:
        g.latents_export(v, e)
    rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, 2)
    assert np.abs(v - rv).max() / np.abs(rv).max() < 1e-6
    assert np.abs(e - re).max() / np.abs(re).max() < 1e-6
