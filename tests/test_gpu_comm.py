"""The multi-partition path behind the C ABI (BASELINE.json configs[3]): after mgn_comm_init every rank calls
mgn_processor_steps_dev / mgn_forward and the halo exchange plus the overlap schedule run INSIDE the library.  One-GPU box:
the ranks are threads or processes that share device 0 and meet over the MGN_COMM_HOST transport (RCCL refuses two ranks on one
device); the RCCL transport itself is driven at world size 1 (same code path: pack, grouped send/recv on the communication
stream, events both ways, phase-split launches)."""
import os
import subprocess
import sys
import threading
from importlib import import_module

import numpy as np
import pytest
import torch

import mgn_amd
import mgn_oracle as orc
from mgn_amd import synth
from util import TOL_15, cfg_dict, engine_for, make_params, rel_max, set_kernel_path

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(P, body):
    """body(rank) in P threads (ctypes releases the GIL inside the library); re-raises the first failure."""
    res, errs = {}, {}

    def work(k):
        try:
            res[k] = body(k)
        except BaseException as ex:   # noqa: BLE001
            errs[k] = ex

    ts = [threading.Thread(target=work, args=(k,)) for k in range(P)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    if errs:
        raise next(iter(errs.values()))
    assert sorted(res) == list(range(P)), "a rank did not finish"
    return [res[k] for k in range(P)]


def _problem(mps=4, nx=40, ny=33, seed=9):
    cfg = cfg_dict(mps=mps)
    pos, cells = synth.grid_mesh(nx, ny, seed)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    rng = np.random.default_rng(1)
    v0 = rng.standard_normal((N, 128)).astype(np.float32)
    e0 = rng.standard_normal((E, 128)).astype(np.float32)
    return cfg, pos, s, r, N, E, ps, v0, e0


@pytest.mark.parametrize("path", [1, 3], ids=["resident", "cooperative"])
@pytest.mark.parametrize("P", [2, 4, 8])
def test_kat7_in_library_schedule_equals_python_driven_bitwise(P, path):
    """KAT-7 through the C entry: P ranks call mgn_processor_steps_dev; the result is BITWISE what the Python-driven twin
    (engine.run_processor_staged + LoopbackExchange) produces -- same kernels, same tile ranges, same order -- and the
    single-partition result up to fp32 summation order."""
    halo = import_module("mgn_amd.halo")
    old = set_kernel_path(path)
    try:
        cfg, pos, s, r, N, E, ps, v0, e0 = _problem()
        single = engine_for(cfg)
        single.set_params(ps)
        single.set_graph(s, r, N)
        v1, e1 = single.processor_steps(v0, e0, 4)
        # Python-driven twin
        stream = torch.cuda.current_stream().cuda_stream
        engs = []
        for k in range(P):
            e = engine_for(cfg, rank=k, nranks=P)
            e.set_stream(stream)
            e.set_params(ps)
            e.set_graph(s, r, N, mesh_pos=pos)
            e.latents_import(v0, e0)
            engs.append(e)
        mgn_amd.run_processor_staged(engs, halo.LoopbackExchange(engs, torch.device("cuda")), 4)
        torch.cuda.synchronize()
        vp, ep = np.zeros((N, 128), np.float32), np.zeros((E, 128), np.float32)
        for g in engs:
            g.latents_export(vp, ep)
            g.close()
        # library-driven
        cid = mgn_amd.Engine.comm_unique_id("host")
        vc, ec = np.zeros((N, 128), np.float32), np.zeros((E, 128), np.float32)
        lock = threading.Lock()

        def body(k):
            e = engine_for(cfg, rank=k, nranks=P, device=0)
            e.set_params(ps)
            e.set_graph(s, r, N, mesh_pos=pos)
            e.comm_init(cid, "host")
            e.latents_import(v0, e0)
            e.processor_steps_dev(4)
            e.synchronize()
            with lock:
                e.latents_export(vc, ec)
            n_halo = e.n_halo
            e.comm_barrier()
            e.close()
            return n_halo

        assert all(n > 0 for n in run_ranks(P, body))
        assert np.array_equal(vc, vp) and np.array_equal(ec, ep)
        assert rel_max(vc, v1) <= 1e-5 and rel_max(ec, e1) <= 1e-5
        rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, 4)
        assert rel_max(vc, rv) <= TOL_15 and rel_max(ec, re) <= TOL_15
    finally:
        set_kernel_path(old)


def test_forward_on_partitions_built_by_the_rank_local_ingest():
    """mgn_partition_nodes + mgn_set_graph_local (every rank hands over only the edges it has an end of) in front of the same forward:
    bitwise the output of handles built by mgn_set_graph from the global lists, and within tolerance of the oracle."""
    cfg = cfg_dict(mps=3)
    pos, cells = synth.grid_mesh(21, 19, 2)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    rng = np.random.default_rng(2)
    nf, ef = rng.standard_normal((N, 9)).astype(np.float32), rng.standard_normal((E, 3)).astype(np.float32)
    ref = orc.forward(ps, cfg, nf, ef, s, r)
    owner = mgn_amd.Engine.partition_nodes(N, 3, mesh_pos=pos)
    res = {}
    for local in (False, True):
        cid = mgn_amd.Engine.comm_unique_id("host")

        def body(k):
            e = engine_for(cfg, rank=k, nranks=3, device=0)
            e.set_params(ps)
            if local:
                e.set_graph_local(s, r, N, owner)
            else:
                e.set_graph(s, r, N, mesh_pos=pos)
            e.comm_init(cid, "host")
            out = e.forward(nf, ef)
            e.comm_barrier()
            e.close()
            return out

        res[local] = run_ranks(3, body)
    for a, b in zip(res[False], res[True]):
        assert np.array_equal(a, b)
    assert rel_max(res[True][0], ref) <= TOL_15


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_forward_at_nranks_3_returns_the_complete_output_on_every_rank(dtype):
    """mgn.model(graph, ps, st) (reference src/solve.jl:200) on a mesh cut in three: every rank passes the global FeatureGraph
    arrays, uploads the rows it owns, and gets the complete output."""
    cfg = cfg_dict(mps=3)
    pos, cells = synth.grid_mesh(21, 19, 2)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = make_params(cfg)
    rng = np.random.default_rng(2)
    nf, ef = rng.standard_normal((N, 9)).astype(np.float32), rng.standard_normal((E, 3)).astype(np.float32)
    ref = orc.forward(ps, cfg, nf, ef, s, r)
    cid = mgn_amd.Engine.comm_unique_id("host")

    def body(k):
        e = engine_for(cfg, rank=k, nranks=3, device=0, dtype=dtype)
        e.set_params(ps)
        e.set_graph(s, r, N, mesh_pos=pos)
        e.comm_init(cid, "host")
        out1 = e.forward(nf, ef)
        out2 = e.forward(nf, ef)
        e.comm_barrier()
        e.close()
        return out1, out2

    outs = run_ranks(3, body)
    for o1, o2 in outs:
        assert np.array_equal(o1, outs[0][0]) and np.array_equal(o2, o1)          # complete, identical, repeatable
    if dtype == "f32":
        assert rel_max(outs[0][0], ref) <= TOL_15
    else:
        assert np.linalg.norm(outs[0][0] - ref) / np.linalg.norm(ref) <= 3e-2


def test_two_edge_sets_partitioned_in_library():
    """flag_simple-shaped model (mesh + world edges, BASELINE.json configs[2]) cut in two: a halo row carries both sets' P rows
    (the receive buffer is unpacked per set instead of landing in P directly)."""
    mf = synth.mesh_flag(1234, 14, 12, radius=0.13)
    N, E1, E2 = mf["mesh_pos"].shape[0], mf["s"].size, mf["s2"].size
    cfg = dict(Fn=12, Fe=7, O=3, L=128, hidden_layers=2, mps=3)
    ps = orc.init_params(12, 7, 3, 128, 2, 3, 5, 0.1, Fe2=4)
    rng = np.random.default_rng(4)
    v0 = rng.standard_normal((N, 128)).astype(np.float32)
    e0 = rng.standard_normal((E1, 128)).astype(np.float32)
    w0 = rng.standard_normal((E2, 128)).astype(np.float32)

    def make(**kw):
        e = mgn_amd.Engine(12, 7, 3, 128, 2, 3, Fe2=4, **kw)
        e.set_params(ps)
        e.set_graph(mf["s"], mf["r"], N, mesh_pos=mf["world_pos"])
        e.set_edge_set(1, mf["s2"], mf["r2"])
        e.latents_import(v0, e0)
        e.edge_latents_import(1, w0)
        return e

    one = make()
    one.processor_steps_dev(3)
    v1, e1 = one.latents_export()
    w1 = one.edge_latents_export(1)
    one.close()
    cid = mgn_amd.Engine.comm_unique_id("host")
    vc, ec, wc = np.zeros_like(v1), np.zeros_like(e1), np.zeros_like(w1)
    lock = threading.Lock()

    def body(k):
        e = make(rank=k, nranks=2, device=0)
        e.comm_init(cid, "host")
        e.processor_steps_dev(3)
        with lock:
            e.latents_export(vc, ec)
            e.edge_latents_export(1, wc)
        e.comm_barrier()
        e.close()
        return True

    run_ranks(2, body)
    assert rel_max(vc, v1) <= 1e-5 and rel_max(ec, e1) <= 1e-5 and rel_max(wc, w1) <= 1e-5
    # ... and the whole model: mgn_forward with both edge sets on the partitioned mesh against the oracle
    rng = np.random.default_rng(8)
    nf = rng.standard_normal((N, 12)).astype(np.float32)
    ref = orc.forward(ps, dict(cfg, Fe2=4), nf, mf["ef"], mf["s"], mf["r"], set2=(mf["ef2"], mf["s2"], mf["r2"]))
    cid2 = mgn_amd.Engine.comm_unique_id("host")

    def fwd(k):
        e = make(rank=k, nranks=2, device=0)
        e.set_edge_features(1, mf["ef2"])
        e.comm_init(cid2, "host")
        out = e.forward(nf, mf["ef"])
        e.comm_barrier()
        e.close()
        return out

    o = run_ranks(2, fwd)
    assert np.array_equal(o[0], o[1]) and rel_max(o[0], ref) <= TOL_15


_RCCL_SELFTEST = r"""
import os, sys
sys.path.insert(0, %(root)r)
os.environ["MGN_FORCE_STAGED"] = "1"
import numpy as np, mgn_amd
from mgn_amd import synth
pos, cells = synth.grid_mesh(64, 48, 3)
s, r = synth.cells_to_edges(cells)
N = pos.shape[0]
ps = (np.random.default_rng(0).standard_normal(mgn_amd.Engine(9, 3, 2, 128, 2, 3, device=mgn_amd.MGN_DEVICE_NONE).param_count) * 0.05).astype(np.float32)
a = mgn_amd.Engine(9, 3, 2, 128, 2, 3, device=0)
a.set_params(ps); a.set_graph(s, r, N); a.latents_randn(5); a.processor_steps_dev(3); ca = a.latents_checksum()
b = mgn_amd.Engine(9, 3, 2, 128, 2, 3, device=0)
b.set_params(ps); b.set_graph(s, r, N)
b.comm_init(mgn_amd.Engine.comm_unique_id("rccl"), "rccl")       # RCCL communicator of one rank
b.latents_randn(5); b.processor_steps_dev(3); cb = b.latents_checksum()
assert b.comm_allreduce([3.0, 4.0], "sum").tolist() == [3.0, 4.0] and b.comm_allreduce([7.0], "max")[0] == 7.0
b.comm_barrier()
out = b.forward(np.ones((N, 9), np.float32), np.ones((s.size, 3), np.float32))
# (the staged schedule projects P / Q in a launch of its own where the plain pass of a mesh this small fuses the projection into the node
#  kernel: the same arithmetic on another kernel family -- equal to rounding, not bitwise, since the edge kernel of this size class changed)
assert all(abs(ca[k] - cb[k]) <= 1e-6 * max(abs(ca[k]), 1.0) for k in ca), (ca, cb)
assert np.isfinite(out).all()
b.close(); a.close()
print("RCCL-OK")
"""


def test_rccl_transport_at_world_size_1():
    """The production transport: RCCL bound at run time, communicator of one rank, staged schedule forced
    (MGN_FORCE_STAGED): checksums equal the plain single-partition pass (to rounding: the two schedules may run the node side on different
    kernel families)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", _RCCL_SELFTEST % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "RCCL-OK" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


_PROC_RANK = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import numpy as np, mgn_amd
from mgn_amd import synth
rank, world, path, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
import mgn_oracle as orc
cfg = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=2, mps=3)
pos, cells = synth.grid_mesh(37, 29, 2)
s, r = synth.cells_to_edges(cells)
N, E = pos.shape[0], s.size
ps = orc.init_params(9, 3, 2, 128, 2, 3, 11, 0.1)
rng = np.random.default_rng(3)
v0 = rng.standard_normal((N, 128)).astype(np.float32); e0 = rng.standard_normal((E, 128)).astype(np.float32)
eng = mgn_amd.Engine(9, 3, 2, 128, 2, 3, rank=rank, nranks=world, device=0, dtype=dtype)
eng.set_params(ps); eng.set_graph(s, r, N, mesh_pos=pos)
eng.comm_init_file(os.path.join(path, "comm.id"), "host")
eng.latents_import(v0, e0)
eng.processor_steps_dev(3)
v, e = eng.latents_export()
np.savez(os.path.join(path, "rank%%d.npz" %% rank), v=v, e=e, n_halo=eng.n_halo)
eng.comm_barrier(); eng.close()
"""


@pytest.mark.parametrize("world,dtype", [(2, "f32"), (3, "f32"), (2, "bf16")])
def test_separate_processes_one_partition_each(world, dtype, tmp_path):
    """As close as a one-GPU box gets to `bench.py --gpus N`: separate processes, the real engine in each, nothing but the C
    entry points (file bootstrap, mgn_processor_steps_dev)."""
    env = dict(os.environ)
    procs = [subprocess.Popen([sys.executable, "-c", _PROC_RANK % dict(root=ROOT), str(k), str(world), str(tmp_path), dtype],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(world)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-2000:] for o in outs)
    cfg = dict(Fn=9, Fe=3, O=2, L=128, hidden_layers=2, mps=3)
    pos, cells = synth.grid_mesh(37, 29, 2)
    s, r = synth.cells_to_edges(cells)
    N, E = pos.shape[0], s.size
    ps = orc.init_params(9, 3, 2, 128, 2, 3, 11, 0.1)
    rng = np.random.default_rng(3)
    v0 = rng.standard_normal((N, 128)).astype(np.float32)
    e0 = rng.standard_normal((E, 128)).astype(np.float32)
    rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, 3)
    v, e = np.zeros((N, 128), np.float32), np.zeros((E, 128), np.float32)
    for k in range(world):
        d = np.load(tmp_path / f"rank{k}.npz")
        assert d["n_halo"] > 0
        v += d["v"]            # owned rows are disjoint, the others are zero: the sum merges the partitions
        e += d["e"]
    if dtype == "f32":
        assert rel_max(v, rv) <= TOL_15 and rel_max(e, re) <= TOL_15
    else:
        l2 = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        assert l2(v, rv) <= 3e-2 and l2(e, re) <= 3e-2


def test_full_size_1m_eight_partitions_in_library():
    """BASELINE.json configs[3] at its real size and partition count (N = 1 000 000, E = 5 992 002, eight edge-cut
    partitions), all eight on the one GPU of the box: checksums of the in-library staged pass equal the single-partition pass
    (partition invariance; latents are keyed by global ids) and two runs are bitwise equal."""
    cfg = cfg_dict(mps=2)
    pos, s, r = synth.mesh_1m(1234)
    N, E = pos.shape[0], s.size
    assert (N, E) == (1000000, 5992002)
    ps = make_params(cfg, jitter=0.05)
    one = engine_for(cfg)
    one.set_params(ps)
    one.set_graph(s, r, N)
    one.latents_randn(77)
    one.processor_steps_dev(2)
    a = one.latents_checksum()
    one.close()

    def run():
        cid = mgn_amd.Engine.comm_unique_id("host")

        def body(k):
            e = engine_for(cfg, rank=k, nranks=8, device=0)
            e.set_params(ps)
            e.set_graph(s, r, N, mesh_pos=pos)
            e.comm_init(cid, "host")
            e.latents_randn(77)
            e.processor_steps_dev(2)
            c = e.latents_checksum()
            tot = e.comm_allreduce([c["sum_v"], c["sum_e"], c["sumsq_v"], c["sumsq_e"], float(e.n_own), float(e.e_local)])
            e.comm_barrier()
            e.close()
            return tot.tolist()

        res = run_ranks(8, body)
        assert all(x == res[0] for x in res)          # the reduction gives every rank the same bits
        return res[0]

    c = run()
    assert c == run()
    assert c[4] == N and c[5] == E
    for k, v in zip(("sum_v", "sum_e", "sumsq_v", "sumsq_e"), c):
        assert abs(v - a[k]) <= 2e-6 * max(abs(a[k]), 1.0) + 1e-3 * (k.startswith("sum_")), (k, a[k], v)


def test_bench_two_ranks_under_torchrun_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, rendezvous store for the communicator id),
    with both ranks on device 0 of this one-GPU box: RCCL refuses two ranks on one device, every rank reports that through the
    store and all fall back to the library's shared-memory transport -- the timed loop (mgn_processor_steps_dev at nranks = 2)
    and the JSON contract are what a real 2-GPU run executes."""
    import json
    env = dict(os.environ, MGN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 300)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nx", "200"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-1500:] + res.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["latents_finite"] and d["scaling"] == "strong"
    assert d["per_rank"]["n_halo"] > 0 and "transport" in d["config"]["partition"]


def test_bench_two_gpus_without_a_launcher_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher (WORLD_SIZE unset): bench.py starts torch.distributed.run itself, as a child
    process and before it has touched the GPU, and relays the JSON line and the exit code.  Both ranks sit on
    device 0 here (MGN_BENCH_ONE_GPU), so the transport decision -- taken before any rank calls RCCL -- is the shared-memory one."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MGN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nx", "200"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-1500:] + res.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["latents_finite"]
    assert d["transport"].startswith("host") and len(d["per_rank"]["graph_setup_s"]) == 2 and d["median"]["value"] > 0


@pytest.mark.parametrize("P", [2, 3])
def test_rollout_and_ode_step_on_a_partitioned_mesh(P):
    """rollout (reference src/solve.jl:42-68) and ode_step (:188-219) at nranks > 1: every rank passes the global arrays, integrates
    the rows it owns (inflow overwrite, Euler / adaptive Tsit5 with a rank-reduced error norm) and returns the complete solution --
    equal on all ranks and equal to the single-partition run up to fp32 summation order; Euler also against GOLD-D."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "gold_d_rollout.npz"))
    ps = orc.init_params(9, 3, 2, int(g["L"]), 2, int(g["mps"]), seed=int(g["seed"]), ln_jitter=float(g["jitter"]))
    cfg = cfg_dict(L=int(g["L"]), mps=int(g["mps"]))
    onehot = orc.one_hot(g["node_type"], 7, 0).astype(np.float32)
    dt = float(g["dt"])
    N = g["x0"].shape[0]

    def setup(e):
        e.set_params(ps)
        e.set_graph(g["senders"], g["receivers"], N, mesh_pos=g["mesh_pos"])
        e.set_norms(node=(g["node_scale"], g["node_shift"]), edge=(g["edge_scale"], g["edge_shift"]), out=(g["out_scale"], g["out_shift"]))

    def run(e):
        kw = dict(val_mask=g["val_mask"], inflow_mask=g["inflow_mask"][:, 0], inflow_data=g["gt"], inflow_rule="tolerant")
        eu, st = e.rollout("Euler", g["x0"], onehot, g["ef_raw"], 0.0, 10 * dt, dt, 11, dt=dt, **kw)
        ts, st2 = e.rollout("Tsit5", g["x0"], onehot, g["ef_raw"], 0.0, 10 * dt, dt, 11, abstol=1e-6, reltol=1e-3, **kw)
        f1 = e.ode_step(g["x0"].astype(np.float32), onehot, g["ef_raw"], g["val_mask"])
        e.set_static(onehot, g["ef_raw"], g["val_mask"])
        f2 = e.ode_step(g["x0"].astype(np.float32))
        return eu, ts, f1, f2, st["n_rhs"], st2["n_accept"]

    one = engine_for(cfg)
    setup(one)
    ref = run(one)
    one.close()
    cid = mgn_amd.Engine.comm_unique_id("host")

    def body(k):
        e = engine_for(cfg, rank=k, nranks=P, device=0)
        setup(e)
        e.comm_init(cid, "host")
        out = run(e)
        e.comm_barrier()
        e.close()
        return out

    outs = run_ranks(P, body)
    for o in outs:
        for a, b in zip(o[:4], outs[0][:4]):
            assert np.array_equal(a, b)                       # complete and identical on every rank
        assert o[4] == 10 and o[5] == ref[5]
    eu, ts, f1, f2 = outs[0][:4]
    assert np.array_equal(f1, f2)
    assert rel_max(eu, ref[0]) <= 1e-5 and rel_max(ts, ref[1]) <= 1e-5 and rel_max(f1, ref[2]) <= 1e-5
    assert np.linalg.norm(eu - g["xs"]) / np.linalg.norm(g["xs"]) <= 1e-3


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("P", [2, 4])
def test_kat7_split_path_partitions_against_the_oracle(P, dtype):
    """KAT-7 where the DEFAULT kernels run it (kernel path 0): a 150 x 150 slice of the M-1M generator (22 500 nodes, 133 k edges) in P
    edge-cut partitions is large enough for the ring kernel of the split path (fp32) / the persistent bf16 kernels to take every
    launch, interior and boundary tile ranges alike; the result is held to the float64 oracle (not only to the single-partition run)."""
    from util import set_kernel_path
    old = set_kernel_path(0)
    try:
        cfg = cfg_dict(mps=4)
        pos, s, r = synth.mesh_1m(1234, 150, 150)
        N, E = pos.shape[0], s.size
        ps = make_params(cfg, jitter=0.05)
        rng = np.random.default_rng(5)
        v0 = rng.standard_normal((N, 128)).astype(np.float32)
        e0 = rng.standard_normal((E, 128)).astype(np.float32)
        kw = dict(dtype="bf16") if dtype == "bf16" else {}
        single = engine_for(cfg, **kw)
        single.set_params(ps)
        single.set_graph(s, r, N)
        v1, e1 = single.processor_steps(v0, e0, 4)
        cid = mgn_amd.Engine.comm_unique_id("host")
        vc, ec = np.zeros((N, 128), np.float32), np.zeros((E, 128), np.float32)
        lock = threading.Lock()

        def body(k):
            e = engine_for(cfg, rank=k, nranks=P, device=0, **kw)
            e.set_params(ps)
            e.set_graph(s, r, N, mesh_pos=pos)
            e.comm_init(cid, "host")
            e.latents_import(v0, e0)
            e.processor_steps_dev(4)
            e.synchronize()
            with lock:
                e.latents_export(vc, ec)
            n_halo = e.n_halo
            e.comm_barrier()
            e.close()
            return n_halo

        assert all(n > 0 for n in run_ranks(P, body))
        rv, re = orc.processor_steps(ps, cfg, v0, e0, s, r, 4)
        if dtype == "f32":
            assert rel_max(vc, v1) <= 1e-5 and rel_max(ec, e1) <= 1e-5
            assert rel_max(vc, rv) <= TOL_15 and rel_max(ec, re) <= TOL_15
        else:       # bf16 storage: the band of tests/test_gpu_bf16.py (relative L2 against the float64 oracle)
            for a, b in ((vc, rv), (ec, re), (v1, rv), (e1, re)):
                assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 3e-2
    finally:
        set_kernel_path(old)
