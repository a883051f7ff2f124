"""The communicator behind the C ABI (mgn_comm_*, SURVEY.md 8b "the engine owns its communicators", 8e), CPU side: host-only
handles (MGN_DEVICE_NONE: partitioner + halo lists, no compute) of a partitioned mesh meet over the MGN_COMM_HOST transport
(POSIX shared memory) in threads and in separate processes and exchange per-node rows along the halo lists.  Integer-valued
rows, so every check is exact.  The device side of the same entry points is tests/test_gpu_comm.py."""
import multiprocessing as mp
import os
import sys
import threading
import time

import numpy as np
import pytest

import mgn_amd
from mgn_amd import synth
from mgn_amd.engine import MgnError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mesh():
    pos, cells = synth.grid_mesh(37, 29, 2)
    s, r = synth.cells_to_edges(cells)
    return pos, s, r


def _rank_body(rank, P, cid, pos, s, r, rounds=3):
    e = mgn_amd.Engine(9, 3, 2, 128, 2, 3, rank=rank, nranks=P, device=mgn_amd.MGN_DEVICE_NONE)
    e.set_graph(s, r, pos.shape[0], mesh_pos=pos)
    e.comm_init(cid, "host")
    own = e.owned_nodes()
    ok = True
    for it in range(rounds):          # growing rows: the outboxes are re-created under a new generation
        W = 2 + 7 * it
        rows = np.stack([(own * (k + 1) + it).astype(np.float32) for k in range(W)], 1)
        halo = e.halo_exchange_host(rows)
        hn = e.halo_nodes()
        want = np.stack([(hn * (k + 1) + it).astype(np.float32) for k in range(W)], 1)
        ok = ok and np.array_equal(halo, want)
    tot = e.comm_allreduce([float(e.n_own), float(rank + 1)], "sum")
    mx = e.comm_allreduce([float(rank)], "max")
    e.comm_barrier()
    n_halo = e.n_halo
    e.comm_destroy()
    e.close()
    return ok, tot.tolist(), mx.tolist(), n_halo


@pytest.mark.parametrize("P", [2, 3, 8])
def test_host_transport_threads(lib_built, P):
    pos, s, r = _mesh()
    cid = mgn_amd.Engine.comm_unique_id("host")
    res = {}

    def work(rank):
        res[rank] = _rank_body(rank, P, cid, pos, s, r)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(P)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert sorted(res) == list(range(P))
    for k in range(P):
        ok, tot, mx, n_halo = res[k]
        assert ok and n_halo > 0
        assert tot == [float(pos.shape[0]), P * (P + 1) / 2.0] and mx == [P - 1.0]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("mgn_")]      # nothing left behind


def _proc_main(rank, P, path, q):
    sys.path.insert(0, ROOT)
    pos, s, r = _mesh()
    e = mgn_amd.Engine(9, 3, 2, 128, 2, 3, rank=rank, nranks=P, device=mgn_amd.MGN_DEVICE_NONE)
    e.set_graph(s, r, pos.shape[0], mesh_pos=pos)
    e.comm_init_file(path, "host")          # bootstrap through a file: rank 0 writes the id, the others wait for it
    own = e.owned_nodes()
    halo = e.halo_exchange_host(own.astype(np.float32)[:, None])
    q.put((rank, bool(np.array_equal(halo[:, 0], e.halo_nodes().astype(np.float32))), float(e.comm_allreduce([1.0])[0])))
    e.comm_barrier()
    e.close()


def test_host_transport_processes_with_file_bootstrap(lib_built, tmp_path):
    P = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = str(tmp_path / "comm.id")
    procs = [ctx.Process(target=_proc_main, args=(k, P, path, q)) for k in range(P)]
    [p.start() for p in procs]
    got = sorted(q.get(timeout=120) for _ in range(P))
    [p.join(60) for p in procs]
    assert got == [(k, True, float(P)) for k in range(P)]
    assert all(p.exitcode == 0 for p in procs)


def test_file_bootstrap_ignores_a_stale_id_file_and_cleans_up(lib_built, tmp_path):
    """a second run with the same path (or a rerun after a crash) must not pick up the previous run's id.  The path is
    pre-seeded with a well-formed but stale record, a stale go file and a stale ack; the run must complete with rank 0's fresh id
    (the nonce handshake), twice in a row, and leave no file behind."""
    P = 3
    ctx = mp.get_context("spawn")
    path = str(tmp_path / "comm.id")
    stale = bytes(128) + (12345).to_bytes(8, "little")
    for rnd in range(2):
        with open(path, "wb") as f:
            f.write(stale)
        with open(path + ".go", "wb") as f:
            f.write((12345).to_bytes(8, "little"))
        with open(path + ".ack.1", "wb") as f:
            f.write((12345).to_bytes(8, "little"))
        q = ctx.Queue()
        procs = [ctx.Process(target=_proc_main, args=(k, P, path, q)) for k in range(P)]
        # the other ranks first: they find the stale files before rank 0 has removed them
        [p.start() for p in procs[1:]]
        time.sleep(0.3)
        procs[0].start()
        got = sorted(q.get(timeout=120) for _ in range(P))
        [p.join(60) for p in procs]
        assert got == [(k, True, float(P)) for k in range(P)], rnd
        assert all(p.exitcode == 0 for p in procs)
        assert not [f for f in os.listdir(tmp_path) if f.startswith("comm.id")], os.listdir(tmp_path)


def test_comm_errors(lib_built, monkeypatch):
    pos, s, r = _mesh()
    e = mgn_amd.Engine(9, 3, 2, 128, 2, 3, rank=0, nranks=2, device=mgn_amd.MGN_DEVICE_NONE)
    e.set_graph(s, r, pos.shape[0], mesh_pos=pos)
    with pytest.raises(MgnError) as ei:     # exchange before mgn_comm_init
        e.halo_exchange_host(np.zeros((e.n_own, 1), np.float32))
    assert ei.value.code == mgn_amd.MGN_E_RCCL
    with pytest.raises(ValueError):
        e.comm_init(b"short", "host")
    with pytest.raises(MgnError) as ei:     # an id that was not made for this transport
        e.comm_init(bytes(128), "host")
    assert ei.value.code == mgn_amd.MGN_E_RCCL
    with pytest.raises(MgnError) as ei:     # RCCL needs a device handle
        e.comm_init(bytes(128), "rccl")
    assert ei.value.code == mgn_amd.MGN_E_RCCL
    monkeypatch.setenv("MGN_COMM_TIMEOUT_S", "0.5")
    with pytest.raises(MgnError) as ei:     # the peer never arrives: a time-out, not a hang
        e.comm_init(mgn_amd.Engine.comm_unique_id("host"), "host")
    assert ei.value.code == mgn_amd.MGN_E_RCCL and "timed out" in str(ei.value)
    e.close()
    for f in os.listdir("/dev/shm"):
        if f.startswith("mgn_"):
            os.unlink(os.path.join("/dev/shm", f))


def test_single_rank_comm_is_a_no_op_exchange(lib_built):
    pos, s, r = _mesh()
    e = mgn_amd.Engine(9, 3, 2, 128, 2, 3, device=mgn_amd.MGN_DEVICE_NONE)
    e.set_graph(s, r, pos.shape[0])
    e.comm_init(mgn_amd.Engine.comm_unique_id("host"), "host")
    assert e.halo_exchange_host(np.ones((e.n_own, 3), np.float32)).shape == (0, 3)
    assert e.comm_allreduce([2.5], "max")[0] == 2.5
    e.close()
