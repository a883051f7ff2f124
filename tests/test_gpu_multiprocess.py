"""Several PROCESSES, one partition each, the real engine in every one of them -- on the single GPU of the test box: the
processes share device 0 and exchange halo rows through gloo (HostStagedExchange).  RCCL refuses two ranks on one device,
so this is as close as a 1-GPU box gets to `bench.py --gpus N`: same partitioner, same staged driver with the split edge
step, same pack / unpack kernels, a real process group.  Run on the MI355X box with `-m gpu`."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, tmp, dtype):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")                      # torch initialises its runtime before the engine's first HIP call
    import mgn_amd
    import mgn_oracle as orc
    from importlib import import_module
    halo = import_module("mgn_amd.halo")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=2, mps=3)
    pos, cells = mgn_amd.synth.grid_mesh(37, 29, 2)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = orc.init_params(9, 3, 2, 128, 2, 3, 11, 0.1)
    rng = np.random.default_rng(3)
    v0 = rng.standard_normal((N, 128)).astype(np.float32)
    e0 = rng.standard_normal((E, 128)).astype(np.float32)
    eng = mgn_amd.Engine(9, 3, 2, 128, 2, 3, rank=rank, nranks=world, device=0, dtype=dtype)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_params(ps)
    eng.set_graph(s, r, N, mesh_pos=pos)
    eng.latents_import(v0, e0)
    ex = halo.HostStagedExchange(eng, torch.device("cuda", 0))
    mgn_amd.run_processor_staged([eng], ex, 3)
    torch.cuda.synchronize()
    v, e = np.zeros((N, 128), np.float32), np.zeros((E, 128), np.float32)
    eng.latents_export(v, e)
    tv, te = torch.from_numpy(v), torch.from_numpy(e)
    dist.all_reduce(tv)                                 # owned rows are disjoint: the sum merges the partitions
    dist.all_reduce(te)
    if rank == 0:
        rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, 3)
        l2 = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        np.savez(os.path.join(tmp, "res.npz"), dv=np.abs(tv.numpy() - rv).max() / np.abs(rv).max(),
                 de=np.abs(te.numpy() - re).max() / np.abs(re).max(), lv=l2(tv.numpy(), rv), le=l2(te.numpy(), re), n_halo=eng.n_halo)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype", [(2, "f32"), (3, "f32"), (2, "bf16")])
def test_real_engines_in_separate_processes(world, dtype):
    port = 29700 + (os.getpid() % 500) + 7 * world + (3 if dtype == "bf16" else 0)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, port, tmp, dtype), nprocs=world, join=True)
        res = np.load(os.path.join(tmp, "res.npz"))
        assert res["n_halo"] > 0
        if dtype == "f32":
            assert res["dv"] <= 1e-4 and res["de"] <= 1e-4, (res["dv"], res["de"])
        else:
            assert res["lv"] <= 3e-2 and res["le"] <= 3e-2, (res["lv"], res["le"])


def test_engine_first_then_torch_in_one_process():
    """The loader imports torch before it dlopens the library (PyTorch-ROCm bundles its own HIP runtime; whichever copy
    initialises second finds no GPU): a program that uses the engine FIRST and a device tensor afterwards has to work."""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, mgn_amd\n"
        "pos, cells = mgn_amd.synth.grid_mesh(9, 7, 1); s, r = mgn_amd.synth.cells_to_edges(cells)\n"
        "eng = mgn_amd.Engine(9, 3, 2, 32, 2, 1); eng.set_params(np.zeros(eng.param_count, np.float32)); eng.set_graph(s, r, pos.shape[0])\n"
        "assert 'torch' in sys.modules          # the loader did it\n"
        "out = eng.forward(np.zeros((pos.shape[0], 9), np.float32), np.zeros((s.size, 3), np.float32))\n"
        "import torch\n"
        "t = torch.ones(4, device='cuda') * 2\n"
        "assert float(t.sum()) == 8.0 and np.isfinite(out).all()\n"
        "print('OK')\n" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "OK" in res.stdout, res.stderr[-2000:]
