#!/usr/bin/env python3
"""bench.py -- processor-step throughput of the MI355X-native MeshGraphNets engine.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU (RCCL), RANK/LOCAL_RANK/WORLD_SIZE from env.
  One "step" = one pass of the hot path over the mesh = `mps` (15) message-passing steps of the processor
  (mgn_processor_steps_dev; at N > 1 the staged edge/node kernels + RCCL halo all-to-all-v between them).
  W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + device synchronise on both
  sides; time = MAX over ranks; rank 0 prints ONE JSON line.

Metric (BASELINE.json): processor-step edges/s (+ nodes/s) = E * mps * K / time  -- whole job, all GPUs.
Workload at N = 1: M-1M, the 1000x1000 jittered-grid triangulation (N = 1 000 000 nodes, E = 5 992 002
directed edges, L = 128, 15 steps, fp32) -- BASELINE.json configs[3], the config the multi-GPU metric is
quoted on; it fits one GPU (5 GB).  configs[1] (cylinder_flow-sized M-cyl) is timed too and reported in the
same line under "secondary" (it is latency-bound: ~12k edges).  At N > 1 the SAME mesh is edge-cut over
the N GPUs ("scaling": "strong").  Latents are N(0,1) generated on device, weights Glorot-uniform
(random init: no datasets/checkpoints exist offline) -- "data": "synthetic".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (the headline figures with 2:1 sparsity are not used)
PEAK_HBM_GBPS = 8000.0         # same guide: HBM3E ~8 TB/s
L, MPS, FN, FE, O = 128, 15, 9, 3, 2


def glorot_params(seed=1234, Fn=None, Fe=None, O_=None, Fe2=None):
    """Glorot-uniform W, zero b, gamma = 1, beta = 0 in MGN-spec packed order (include/mgn_hip.h).
    Fe2: a second edge set (its encoder and per-step edge MLP; node MLP input 3L)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    chunks = []

    def mlp(n_in, n_out, ln):
        dims = [n_in, L, L, n_out]
        for i in range(3):
            lim = (6.0 / (dims[i] + dims[i + 1])) ** 0.5
            chunks.append(rng.uniform(-lim, lim, size=dims[i] * dims[i + 1]).astype(np.float32))
            chunks.append(np.zeros(dims[i + 1], np.float32))
        if ln:
            chunks.append(np.ones(n_out, np.float32))
            chunks.append(np.zeros(n_out, np.float32))

    mlp(Fn or FN, L, True)
    mlp(Fe or FE, L, True)
    if Fe2:
        mlp(Fe2, L, True)
    for _ in range(MPS):
        mlp(3 * L, L, True)
        if Fe2:
            mlp(3 * L, L, True)
        mlp((3 if Fe2 else 2) * L, L, True)
    mlp(L, O_ or O, False)
    return np.concatenate(chunks)


def flops_algorithmic(E, N):
    """GEMM flops of the un-factored processor step."""
    return 163840.0 * E + 131072.0 * N


def flops_edge_kernel(E):
    """MFMA flops the edge kernel executes: 3 chunks of L x L per edge (layer 1 is factored)."""
    return 2.0 * 3 * L * L * E


def flops_node_kernel(N, project):
    return 2.0 * (4 + (2 if project else 0)) * L * L * N


def cpu_baseline(ps, budget_s=20.0):
    """oracle/mgn_ref.c (fp32 C restatement, OpenMP) timed on this box's host cores on a bounded
    sample of the same workload: processor steps on a 300x300 slice of the M-1M generator."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mgn_ref
    import mgn_amd
    pos, cells = mgn_amd.synth.grid_mesh(300, 300, 1234)
    s, r = mgn_amd.synth.cells_to_edges(cells)
    n, e = pos.shape[0], s.size
    rng = np.random.default_rng(1234)
    v = rng.standard_normal((n, L)).astype(np.float32)
    el = rng.standard_normal((e, L)).astype(np.float32)
    cfg = dict(Fn=FN, Fe=FE, O=O, L=L, mps=MPS)
    t0 = time.time()
    steps = 0
    while steps < MPS and (time.time() - t0 < budget_s or steps == 0):
        v, el = mgn_ref.processor_steps(ps, cfg, v, el, s, r, 1)
        steps += 1
    dt = time.time() - t0
    return dict(value=e * steps / dt, unit="edges/s", nodes_per_s=n * steps / dt, cores=mgn_ref.num_threads(),
                kind="port", sample=f"per-edge rate on a SLICE, not on the headline mesh: {steps} processor step(s), L=128 fp32, on a 300x300 "
                f"slice of the M-1M generator (N={n}, E={e}); oracle/mgn_ref.c (the C restatement of MGN-spec, not the Julia path) with OpenMP "
                f"on all host cores; {dt:.1f} s",
                host_cpus=os.cpu_count())


def committed_traffic(prefixes, dtype="f32"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN/pmc_summary_bench_1m[_bf16].json, newest round first; separate --pmc FETCH_SIZE / WRITE_SIZE passes,
    gfx950 x2 read correction as prescribed by the guide's HBM section).  `prefixes`: kernel-name prefixes as they appear in the
    profile keys (the key of a round is matched by prefix, so that template arguments added later do not hide it).
    (None, None) when no profile is committed."""
    names = ["pmc_summary_bench_1m.json", "pmc_summary_bench_1m_fp32_mfma.json"] if dtype == "f32" else ["pmc_summary_bench_1m_bf16.json"]
    pdir = os.path.join(ROOT, "profiles")
    try:
        rounds = sorted((d for d in os.listdir(pdir) if d.startswith("r") and d[1:].isdigit()), reverse=True)
    except OSError:
        return None, None
    for rd in rounds:
        for name in names:
            try:
                d = json.load(open(os.path.join(pdir, rd, name)))
            except Exception:
                continue
            for k, v in d.items():
                kk = k.replace("void ", "").replace("mgn::", "")
                if any(kk.startswith(p) for p in prefixes) and "derived" in v and "hbm_bytes_per_launch_corrected" in v["derived"]:
                    return v["derived"]["hbm_bytes_per_launch_corrected"], f"profiles/{rd}/{name}"
    return None, None


def scaling_model(ps, pos, s, r, N, E, t_step_1gpu, device, sync, xgmi_link_GBps=153.0, exchange_latency_us=20.0):
    """What 2 / 4 / 8 GPUs should give on this mesh, from ONE GPU (no multi-GPU node is available to the build): for P partitions the share
    of a middle rank of the real RCB partition (real halo and send lists, interior / boundary tile split) is run through the staged
    schedule of mgn_processor_steps_dev -- boundary projection, pack, interior projection + interior edge tiles, unpack, boundary edge
    tiles, node update; engine.run_processor_staged, the Python twin of processor_pass_staged -- with the wire left out (the halo rows
    are not refreshed: timing only).  The wire is then priced from the rank's real receive count at a stated link rate and latency and is
    exposed only where it exceeds the interior work it overlaps with.  Predicted edges/s = E / predicted time per step."""
    import torch
    import mgn_amd
    model = {"assumptions": {"xgmi_link_GBps": xgmi_link_GBps, "exchange_latency_us": exchange_latency_us,
                             "note": "one process per GPU, one xGMI link per peer pair; the grouped ncclSend / ncclRecv of a step run on the communicator's "
                                     "stream while the interior tiles run (csrc/mgn_api.cpp: processor_pass_staged); per-share times measured on one GPU, "
                                     "wire time assumed, not measured"},
             "one_gpu_ms_per_step": t_step_1gpu * 1e3}
    for P in (2, 4, 8):
        rk = P // 2
        eng = mgn_amd.Engine(FN, FE, O, L, 2, MPS, rank=rk, nranks=P, device=device)
        try:
            eng.set_params(ps)
            t0 = time.perf_counter()
            eng.set_graph(s, r, N, mesh_pos=pos)
            t_set = time.perf_counter() - t0
            eng.latents_randn(1234)
            nsend, nhalo, rowf = int(eng.halo_send_index().size), int(eng.n_halo), eng.halo_row_floats
            send = torch.zeros(max(nsend, 1) * rowf, device=f"cuda:{device}")
            recv = torch.zeros(max(nhalo, 1) * rowf, device=f"cuda:{device}")

            class _Exchange:                                   # pack -> (wire: left out) -> unpack
                def start(self_):
                    eng.halo_pack(send.data_ptr())

                def finish(self_):
                    eng.halo_unpack(recv.data_ptr())

            ex = _Exchange()
            for _ in range(3):
                mgn_amd.run_processor_staged([eng], ex, MPS)
            sync()
            k = 6
            t0 = time.perf_counter()
            for _ in range(k):
                mgn_amd.run_processor_staged([eng], ex, MPS)
            sync()
            t_share = (time.perf_counter() - t0) / (k * MPS)
            cost = {}
            for name, fn, ptr in (("pack_us", eng.halo_pack, send.data_ptr()), ("unpack_us", eng.halo_unpack, recv.data_ptr())):
                for _ in range(10):
                    fn(ptr)
                sync()
                t0 = time.perf_counter()
                for _ in range(100):
                    fn(ptr)
                sync()
                cost[name] = (time.perf_counter() - t0) / 100 * 1e6
            tiles_bnd, tiles_all = eng.edge_boundary_tiles(0)
            tiles_int = tiles_all - tiles_bnd
            wire_us = exchange_latency_us + nhalo * rowf * 4 / (xgmi_link_GBps * 1e3) / max(1, min(2, P - 1))   # strips: two neighbours, a link each
            # the exchange overlaps with the interior projection and the interior edge tiles: ~ the interior share of the step
            t_interior = t_share * tiles_int / max(tiles_int + tiles_bnd, 1)
            exposed = max(0.0, wire_us * 1e-6 - t_interior)
            t_pred = t_share + exposed
            model[f"{P}_gpus"] = {"rank": rk, "n_own": int(eng.n_own), "n_halo": nhalo, "send_rows": nsend, "e_local": int(eng.e_local),
                                  "edge_tiles_interior": int(tiles_int), "edge_tiles_boundary": int(tiles_bnd),
                                  "share_ms_per_step_staged": t_share * 1e3, **cost, "wire_us_assumed": wire_us,
                                  "wire_exposed_us": exposed * 1e6, "predicted_ms_per_step": t_pred * 1e3,
                                  "predicted_edges_per_s": E / t_pred, "predicted_speedup": t_step_1gpu / t_pred,
                                  "predicted_efficiency": t_step_1gpu / t_pred / P, "graph_setup_s": t_set}
        finally:
            eng.close()
    return model


def time_clean(eng, steps, warmup, sync):
    """the timed region: nothing but mgn_processor_steps_dev calls between two barrier + synchronise brackets"""
    for _ in range(warmup):
        eng.processor_steps_dev(MPS)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.processor_steps_dev(MPS)
    sync()
    return time.perf_counter() - t0


def time_each(eng, steps, sync):
    """the median of >= 20 individually timed passes (a separate pass: each step has its own bracket)"""
    ts = []
    for _ in range(steps):
        sync()
        t0 = time.perf_counter()
        eng.processor_steps_dev(MPS)
        sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def kernel_split(eng, steps, sync):
    """per-kernel average launch durations: a SEPARATE pass over the same workload with HIP event pairs around every launch
    group, recorded on the stream the kernels are launched on (mgn_profile_enable); not part of the timed region"""
    sync()
    eng.profile_enable(True)
    for _ in range(steps):
        eng.processor_steps_dev(MPS)
    sync()
    prof = eng.profile_read()
    eng.profile_enable(False)
    return prof


def time_single(eng, steps, warmup, sync):
    dt = time_clean(eng, steps, warmup, sync)
    return dt, kernel_split(eng, max(1, min(steps, 3)), sync)


def rccl_loadable():
    """can this process bind an RCCL library (the engine binds librccl at run time, MGN_RCCL_LIB or the loader's search path)"""
    import ctypes
    for cand in (os.environ.get("MGN_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        if not cand:
            continue
        try:
            ctypes.CDLL(cand)
            return True
        except OSError:
            continue
    return False


def comm_bootstrap(eng, rank, world, device_id):
    """One communicator id for all ranks of the job: rank 0 makes it (mgn_comm_unique_id), the launcher's rendezvous store
    (torch.distributed env:// -- plumbing only) hands it to the others; without a store, a file on this node.
    The transport is decided BEFORE any rank touches RCCL (ncclCommInitRank is collective and has no time-out: a fallback
    agreed on afterwards hangs as soon as one rank's init does not fail): every rank publishes (host, device id, can it load
    librccl) through the store; two ranks on one device or a rank without the library put ALL ranks on the library's
    shared-memory transport.  The RCCL init itself runs under a watchdog that ends the process with a message.
    Returns (store, transport)."""
    import socket
    import threading
    try:
        from torch.distributed import rendezvous
        store, _, _ = next(rendezvous("env://", rank, world))
    except Exception as ex:   # noqa: BLE001
        if world > 1 and "MASTER_PORT" not in os.environ:
            raise
        # No store: nothing can be agreed on before the collective init (are the devices distinct? can every rank load librccl?),
        # and ncclCommInitRank has no time-out -- so this branch takes the library's shared-memory transport (one node by
        # construction: the id travels through a file in /tmp), under the same watchdog as the store path.
        path = f"/tmp/mgn_comm_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}.id"
        sys.stderr.write(f"[bench] store rendezvous failed ({ex!r}); file bootstrap {path}, shared-memory transport\n")
        limit = float(os.environ.get("MGN_COMM_TIMEOUT_S", "120"))

        def give_up_file():
            sys.stderr.write(f"[bench] rank {rank}: file bootstrap did not complete within {limit:.0f} s; giving up\n")
            sys.stderr.flush()
            os._exit(3)

        dog = threading.Timer(limit + 10.0, give_up_file)
        dog.daemon = True
        dog.start()
        try:
            eng.comm_init_file(path, "host")
        finally:
            dog.cancel()
        return None, "host (shared memory: no rendezvous store to agree on RCCL)"
    store.set(f"mgn_dev_{rank}", f"{socket.gethostname()}|{device_id}|{int(rccl_loadable())}".encode())
    devs = [bytes(store.get(f"mgn_dev_{q}")).decode().split("|") for q in range(world)]
    distinct = len({(d[0], d[1]) for d in devs}) == world
    loadable = all(d[2] == "1" for d in devs)
    transport = "rccl" if (distinct and loadable) else "host"
    if rank == 0:
        store.set("mgn_comm_id", eng.comm_unique_id(transport))
    cid = bytes(store.get("mgn_comm_id"))
    limit = float(os.environ.get("MGN_COMM_TIMEOUT_S", "120"))

    def give_up():
        sys.stderr.write(f"[bench] rank {rank}: communicator init ({transport}) did not return within {limit:.0f} s; giving up\n")
        sys.stderr.flush()
        os._exit(3)

    dog = threading.Timer(limit, give_up)
    dog.daemon = True
    dog.start()
    try:
        eng.comm_init(cid, transport)
    finally:
        dog.cancel()
    why = "" if transport == "rccl" else (" (shared memory: " + ("several ranks share a device" if not distinct else "librccl not loadable on every rank") + ")")
    return store, transport + why


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process -- before this process has
    imported torch or touched the GPU (no exec after GPU initialisation on these boxes) -- and relay its output and exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    raise SystemExit(res.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)   # >= 20 timed runs after >= 5 warm-ups
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=1000, help="M-1M grid side (default 1000 -> 1M nodes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = the reference's precision (headline); bf16 = BASELINE.json configs[2] precision")
    ap.add_argument("--force-staged", action="store_true",
                    help="drive the staged multi-partition path (RCCL communicator, in-library schedule) even at world size 1 (self-test)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    if args.force_staged:
        os.environ["MGN_FORCE_STAGED"] = "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)                          # (does not return)
    import numpy as np
    import torch
    import mgn_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or without a launcher)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if os.environ.get("MGN_BENCH_ONE_GPU") == "1":   # tests on a one-GPU box: every rank on device 0 (RCCL refuses that: the
        local_rank = 0                                #   agreed fallback to the shared-memory transport is what gets exercised)
    torch.cuda.set_device(local_rank)
    staged = world > 1 or args.force_staged

    ps = glorot_params()
    pos, s, r = mgn_amd.synth.mesh_1m(1234, args.nx, args.nx)
    N, E = pos.shape[0], int(s.size)

    # one handle == one partition on one GPU.  The halo exchange (RCCL grouped send / recv over xGMI) and the overlap schedule
    # run inside mgn_processor_steps_dev: the timed loop below is the same at every N.
    eng = mgn_amd.Engine(FN, FE, O, L, 2, MPS, rank=rank, nranks=world, device=local_rank, dtype=args.dtype)
    eng.set_params(ps)
    t_setup = time.perf_counter()
    eng.set_graph(s, r, N, mesh_pos=pos)     # unsorted COO in: receiver sort / CSR / partition / halo lists are amortised here
    t_setup = time.perf_counter() - t_setup
    try:
        dev_id = str(torch.cuda.get_device_properties(local_rank).uuid)
    except Exception:   # noqa: BLE001
        dev_id = f"index{local_rank}"
    store, transport = comm_bootstrap(eng, rank, world, dev_id) if staged else (None, None)
    eng.latents_randn(1234)

    def barrier_sync():
        eng.synchronize()
        torch.cuda.synchronize()
        if staged:
            eng.comm_barrier()

    barrier_sync()
    dt = time_clean(eng, args.steps, args.warmup, barrier_sync)
    if staged:
        dt = float(eng.comm_allreduce([dt], "max")[0])      # MAX over ranks
    t_med, t_min = time_each(eng, max(args.steps, 3), barrier_sync)
    if staged:
        t_med = float(eng.comm_allreduce([t_med], "max")[0])
        setup_all = eng.comm_allreduce([t_setup if q == rank else 0.0 for q in range(world)], "sum")
    prof = kernel_split(eng, max(1, min(args.steps, 5)), barrier_sync)

    chk = eng.latents_checksum()
    finite = all(np.isfinite(v) for v in chk.values())
    if staged:
        finite = bool(eng.comm_allreduce([0.0 if finite else 1.0], "max")[0] == 0.0)

    if rank == 0:
        bf = args.dtype == "bf16"
        # a split edge step (N > 1: interior tiles while the halo is in flight, boundary tiles after) is two launches
        t_edge = (prof["edge_step"]["avg_ms"] + (prof["edge_boundary"]["avg_ms"] if prof["edge_boundary"]["count"] else 0.0)) * 1e-3
        t_node = prof["node_step"]["avg_ms"] * 1e-3
        e_loc, n_loc = eng.e_local, eng.n_own
        t_step = dt / (args.steps * MPS)
        # MFMA flops the kernels EXECUTE per pass (layer 1 of the edge MLP is factored into per-node P / Q): edge kernel 3 chunks
        # per edge; node side 4 chunks per node per step + 2 for every projection (15 per pass: 14 next-step + the initial one)
        exec_flops_pass = MPS * flops_edge_kernel(E) + (MPS * 4 + MPS * 2) * 2.0 * L * L * N
        node_flops_pass = (MPS * 4 + MPS * 2) * 2.0 * L * L * n_loc
        node_launches = max(prof["node_step"]["count"], 1) / max(1, min(args.steps, 5))
        edge_kernel = "k_edge_bf16" if bf else "k_edge_step<4,2>"
        if bf:
            # bf16 storage: the step is HBM-bound.  Algorithmic bytes of the edge kernel: e latents R + W,
            # two index streams, one pass over the P, Q rows it gathers and the AGG rows it writes
            edge_bytes = 2.0 * L * 2 * e_loc + 8.0 * e_loc + 3.0 * 2.0 * L * n_loc
            ach = edge_bytes / t_edge / 1e9 if t_edge > 0 else 0.0
            traffic, tsrc = committed_traffic(["k_edge_bf16"], "bf16") if (world == 1 and args.nx == 1000) else (None, None)
            roof = {"bound": "hbm", "kernel": "k_edge_bf16 (fused gather + edge MLP + LayerNorm + residual + segmented scatter, bf16 storage)",
                    "achieved": ach, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBPS,
                    "bytes_per_launch": edge_bytes, "bytes_kind": "algorithmic: 2 L (2 E) latents R+W + 8 E indices + 3 x 2 L N (P, Q rows read once, AGG rows written)",
                    "avg_launch_ms": t_edge * 1e3, "launches": prof["edge_step"]["count"], "traffic": traffic, "traffic_source": tsrc}
            bytes_step = 2.0 * L * (2 * E + 2 * N) + 8.0 * E
            roof["processor_step"] = {"bound": "hbm", "algorithmic_bytes_per_step": bytes_step, "achieved": bytes_step / t_step / 1e9,
                                      "peak": PEAK_HBM_GBPS * world, "unit": "GB/s", "frac": bytes_step / t_step / 1e9 / (PEAK_HBM_GBPS * world),
                                      "note": "SURVEY.md 8(d) compulsory bytes at 2-byte storage / wall time per step",
                                      "mfma_TFLOPs_executed": exec_flops_pass / MPS / t_step / 1e12, "mfma_peak_bf16_dense": 2500.0}
        else:
            import ctypes
            lib = mgn_amd.load()
            lib.mgn_debug_fp32_split.restype = ctypes.c_int
            lib.mgn_debug_fp32_split.argtypes = [ctypes.c_int]
            split_mode = lib.mgn_debug_fp32_split(0)            # query ...
            lib.mgn_debug_fp32_split(split_mode)                # ... and restore
            # the roofline is priced by the kernel family the edge launches RAN on (kernels.hip: launch_edge_step decides by size: a
            # small --nx or a small per-rank partition runs the cooperative / fp32-MFMA kernels whatever the global switch says)
            lib.mgn_debug_last_edge_kernel.restype = ctypes.c_int
            lib.mgn_debug_last_edge_kernel.argtypes = []
            fam = lib.mgn_debug_last_edge_kernel()
            FAMILY = {1: "k_edge_step<.., GEN> (general hidden_layers)", 2: "k_edge_coop16m (16-row tiles)", 3: "k_edge_coop (4-wave tiles)",
                      4: "k_edge_step<4,0> (all-streaming)", 7: "k_edge_ring<8>", 8: "k_edge_ring<4>", 9: "k_edge_step<4,2>",
                      12: "k_edge_coop16m (16-row tiles, split path)",
                      13: "k_edge_ring_h<8>", 14: "k_edge_ring_h<4>", 15: "k_edge_coop16m (16-row tiles, two fp16 pieces)",
                      16: "k_edge_ring_hs<8>", 17: "k_edge_ring_hs<4>"}
            # piece products per fp32 product of the family that ran: 6 (three bf16 pieces), 3 (two fp16 pieces), 0 = fp32 MFMA
            products = {7: 6, 8: 6, 12: 6, 13: 3, 14: 3, 15: 3, 16: 3, 17: 3}.get(fam, 0)
            split_mode = split_mode if products else 0
            comp = (1024.0 + 8.0) * e_loc + 3.0 * 512.0 * n_loc
            alg_bytes = (1024.0 + 8.0 + 85.0) * e_loc
            common = {"avg_launch_ms": t_edge * 1e3, "launches": prof["edge_step"]["count"],
                      "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE); algorithmic = 1117 B/edge",
                      "algorithmic_bytes_per_launch": alg_bytes,
                      # what THIS kernel has to move at least, given the factored design: e latents R + W, index streams, and one
                      # pass over the P, Q rows it gathers and the AGG rows it writes (3 x N x 512 B); gather re-reads come on top
                      "kernel_compulsory_bytes_per_launch": comp}
            if products:
                # fp32 storage, every L x L product as `products` exact 16-bit piece products, fp32 accumulation (csrc/split.hip).  Two
                # roofs: the dense 16-bit MFMA peak with the flops the kernel EXECUTES, and HBM with SURVEY 8(d)'s algorithmic bytes;
                # the line's `bound` is whichever floor is the longer time for this kernel.
                kname = FAMILY[fam].split("<")[0].split(" ")[0]
                fl = products * flops_edge_kernel(e_loc)
                ach = fl / t_edge / 1e12 if t_edge > 0 else 0.0
                ach_b = alg_bytes / t_edge / 1e9 if t_edge > 0 else 0.0
                traffic, tsrc = committed_traffic([kname + "<"]) if (world == 1 and args.nx == 1000) else (None, None)
                pieces = ("two fp16 pieces after a power-of-two scaling per weight chunk and per activation row, three piece products"
                          if products == 3 else "three bf16 pieces, six piece products")
                mfma_roof = {"achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MFMA_TFLOPS,
                             "flops_per_launch": fl, "floor_ms": fl / (PEAK_BF16_MFMA_TFLOPS * 1e12) * 1e3,
                             "flops_kind": f"16-bit MFMA flops executed by this kernel: {products} x 98 304 per edge (layer 1 is factored: the "
                                           "v_s / v_r blocks run per node); against the dense bf16 / fp16 peak"}
                hbm_roof = {"achieved": ach_b, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": ach_b / PEAK_HBM_GBPS,
                            "bytes_per_launch": alg_bytes, "floor_ms": alg_bytes / (PEAK_HBM_GBPS * 1e9) * 1e3,
                            "bytes_kind": "SURVEY.md 8(d) algorithmic bytes of the edge step: 1 024 B of e latents R + W + 8 B of indices + "
                                          "512 N / E B of aggregates per edge"}
                hbm_binds = hbm_roof["floor_ms"] > mfma_roof["floor_ms"]
                bind, other = (hbm_roof, mfma_roof) if hbm_binds else (mfma_roof, hbm_roof)
                roof = {"bound": "hbm" if hbm_binds else "mfma",
                        "kernel": kname + f" (fused gather + edge MLP + LayerNorm + residual + segmented scatter; fp32 operands as {pieces}, "
                        "fp32 accumulate)",
                        **{k: bind[k] for k in ("achieved", "peak", "unit", "frac")},
                        "binding_roof": bind, ("mfma_roof" if hbm_binds else "hbm_roof"): other,
                        "bound_note": "the roof whose floor is the longer time for this kernel (both are given): with three piece products the "
                                      "matrix floor (0.71 ms on M-1M) falls below the HBM floor (0.84 ms)" if hbm_binds else
                                      "matrix floor above the HBM floor",
                        "fp32_equivalent_TFLOPs": flops_edge_kernel(e_loc) / t_edge / 1e12 if t_edge > 0 else 0.0,
                        "fp32_equivalent_note": "the same launch counted as the fp32 products it replaces (98 304 flop per edge): what an fp32-MFMA "
                                                "kernel would have to sustain (its peak: 157.3)",
                        "traffic": traffic, "traffic_source": tsrc, **common}
                ex = products * exec_flops_pass / MPS
                bytes_step = 4.0 * L * (2 * E + 2 * N) + 8.0 * E
                step_m = {"executed_flops_per_step": ex, "achieved": ex / t_step / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS * world,
                          "unit": "TFLOP/s", "frac": ex / t_step / 1e12 / (PEAK_BF16_MFMA_TFLOPS * world)}
                step_h = {"algorithmic_bytes_per_step": bytes_step, "achieved": bytes_step / t_step / 1e9, "peak": PEAK_HBM_GBPS * world,
                          "unit": "GB/s", "frac": bytes_step / t_step / 1e9 / (PEAK_HBM_GBPS * world)}
                sb, so_ = (step_h, step_m) if hbm_binds else (step_m, step_h)
                roof["processor_step"] = {
                    "bound": "hbm" if hbm_binds else "mfma", **sb, ("mfma_roof" if hbm_binds else "hbm_roof"): so_,
                    "note": f"whole step: 16-bit MFMA flops executed ({products} x (98 304 E + 196 608 N)) and SURVEY.md 8(d) bytes "
                            "(1 024 (E + N) + 8 E) over the wall time per step",
                    "fp32_equivalent_TFLOPs": exec_flops_pass / MPS / t_step / 1e12,
                    "algorithmic_equivalent_TFLOPs": flops_algorithmic(E, N) / t_step / 1e12,
                    "algorithmic_equivalent_note": "SURVEY.md 8(d) flops of the UN-factored fp32 algorithm (163 840 E + 131 072 N) over the same "
                                                   "time: a speed-up of the design, not a fraction of any peak"}
            else:
                ach = flops_edge_kernel(e_loc) / t_edge / 1e12 if t_edge > 0 else 0.0
                traffic, tsrc = committed_traffic(["k_edge_step<4, 2"]) if (world == 1 and args.nx == 1000) else (None, None)
                roof = {"bound": "mfma", "kernel": FAMILY.get(fam, "k_edge_step<4,2>") + " (fused gather + edge MLP + LayerNorm + residual + segmented scatter; "
                        "fp32 MFMA)",
                        "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                        "flops_per_launch": flops_edge_kernel(e_loc), "flops_kind": "MFMA flops executed by this kernel (98 304 per edge; "
                        "layer 1 is factored so the v_s/v_r blocks run per node in k_node_step)",
                        "traffic": traffic, "traffic_source": tsrc, **common}
                roof["processor_step"] = {
                    "bound": "mfma", "executed_flops_per_step": exec_flops_pass / MPS, "achieved": exec_flops_pass / MPS / t_step / 1e12,
                    "peak": PEAK_F32_MFMA_TFLOPS * world, "unit": "TFLOP/s", "frac": exec_flops_pass / MPS / t_step / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
                    "note": "MFMA flops the kernels execute (edge 98 304 E; node side 196 608 N incl. the P / Q projection) / wall time per step",
                    "algorithmic_equivalent_TFLOPs": flops_algorithmic(E, N) / t_step / 1e12,
                    "algorithmic_equivalent_note": "SURVEY.md 8(d) flops of the UN-factored algorithm (163 840 E + 131 072 N) over the same time: "
                                                   "a speed-up of the factored design, not a fraction of any peak"}
        roof["kernel_timing"] = "separate pass after the timed region: HIP event pairs per launch on the launch stream (mgn_profile_enable)"
        roof["node_side"] = {"avg_launch_ms": t_node * 1e3, "launches": prof["node_step"]["count"],
                             "achieved_TFLOPs_fp32_equivalent": node_flops_pass / max(node_launches, 1) / t_node / 1e12 if t_node > 0 else 0.0,
                             "note": "node MLP + P / Q projection per event group; flops counted as fp32 products (x 6 executed on the split path)"}
        if prof["halo"]["count"]:
            roof["halo_pack_ms"] = prof["halo"]["avg_ms"]
        if bf:
            precision = "bf16 storage + bf16 MFMA"
        elif split_mode and products == 3:
            precision = ("fp32 storage, fp16x2 split MFMA (every fp32 operand = two fp16 pieces after a power-of-two scaling per weight "
                         "chunk / per activation row, three exact piece products per fp32 product on v_mfma_f32_32x32x16_f16), fp32 "
                         "accumulate; held to the fp32 tolerances against the float64 oracle; MGN_SPLIT_F16=0 gives the bf16x3 path "
                         "(`bf16x3_path`), MGN_FP32_SPLIT=0 the fp32-MFMA path (`fp32_mfma_path`)")
        elif split_mode:
            precision = ("fp32 storage, bf16x3 split MFMA (every fp32 operand = three bf16 pieces, six exact piece products per fp32 "
                         "product on v_mfma_f32_32x32x16_bf16), fp32 accumulate; MGN_FP32_SPLIT=0 gives the fp32-MFMA path reported under "
                         "`fp32_mfma_path`")
        else:
            precision = "fp32 storage, fp32 MFMA (v_mfma_f32_32x32x2_f32; MGN_FP32_SPLIT=0)"
        out = {
            "metric": "processor-step edges/s",
            "value": E * MPS * args.steps / dt,
            "unit": "edges/s",
            "nodes_per_s": N * MPS * args.steps / dt,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_processor_step": t_step * 1e3,
            "median": {"ms_per_step": t_med * 1e3, "value": E * MPS / t_med, "min_ms_per_step": t_min * 1e3,
                       "note": f"SURVEY.md 8(d): median of {max(args.steps, 3)} individually bracketed passes (a separate pass; `value` is the "
                               "driver contract's single bracket around all timed steps)"},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"M-1M jittered-grid triangulation {args.nx}x{args.nx}: N={N} nodes, E={E} directed edges, "
                                   f"L=128, hidden_layers=2, {MPS} processor steps per bench step, {precision} "
                                   f"(BASELINE.json configs[3]{' mesh at configs[2] precision' if bf else ''})",
                       "partition": (f"edge-cut RCB over {world} GPU(s), halo exchange inside mgn_processor_steps_dev, transport {transport}"
                                     if world > 1 else "single partition"),
                       "edge_order": "engine re-sorts by receiver once per trajectory (mgn_set_graph)"},
            "roofline": roof,
            "latents_finite": bool(finite),
            "graph_setup_s": t_setup,   # once per trajectory (mgn_set_graph: sort by receiver, CSR, partition, upload), not in `value`
        }
        if world > 1:
            out["transport"] = transport
            out["per_rank"] = {"n_own": n_loc, "e_local": e_loc, "n_halo": eng.n_halo,
                               "graph_setup_s": [float(x) for x in setup_all],
                               "graph_setup_note": "every rank ingests the global edge lists in mgn_set_graph and keeps its partition"}
        if world == 1 and not args.no_secondary and args.dtype == "f32" and split_mode:
            # the same workload on the fp32-MFMA kernels (v_mfma_f32_32x32x2_f32), for the record beside the split-path headline
            lib.mgn_debug_fp32_split(0)
            try:
                engs = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank)
                engs.set_params(ps)
                engs.set_graph(s, r, N)
                engs.latents_randn(1234)
                dts, profs = time_single(engs, 3, 1, barrier_sync)
                ts = dts / (3 * MPS)
                te = profs["edge_step"]["avg_ms"] * 1e-3
                out["fp32_mfma_path"] = {
                    "workload": "same M-1M mesh, fp32 storage, v_mfma_f32_32x32x2_f32 (MGN_FP32_SPLIT=0: the fp32 reference path; round-2 headline)",
                    "ms_per_processor_step": ts * 1e3, "edges_per_s": E / ts, "edge_kernel_ms": te * 1e3,
                    "node_side_ms": profs["node_step"]["avg_ms"],
                    "roofline": {"bound": "mfma", "kernel": "k_edge_step<4,2>", "achieved": flops_edge_kernel(E) / te / 1e12 if te > 0 else 0.0,
                                 "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": flops_edge_kernel(E) / te / 1e12 / PEAK_F32_MFMA_TFLOPS if te > 0 else 0.0}}
                engs.close()
            finally:
                lib.mgn_debug_fp32_split(split_mode)
        if world == 1 and not args.no_secondary and args.dtype == "f32" and split_mode and products == 3:
            # ... and on the three-piece bf16 split (six products): round 3 / 4's headline path
            lib.mgn_debug_split_f16.restype = ctypes.c_int
            lib.mgn_debug_split_f16.argtypes = [ctypes.c_int]
            oldh = lib.mgn_debug_split_f16(0)
            try:
                engs = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank)
                engs.set_params(ps)
                engs.set_graph(s, r, N)
                engs.latents_randn(1234)
                dts, profs = time_single(engs, 3, 1, barrier_sync)
                ts = dts / (3 * MPS)
                te = profs["edge_step"]["avg_ms"] * 1e-3
                out["bf16x3_path"] = {
                    "workload": "same M-1M mesh, fp32 storage, three bf16 pieces / six piece products (MGN_SPLIT_F16=0: k_edge_ring, k_node_split, "
                                "k_project_split; the headline path of rounds 3 and 4)",
                    "ms_per_processor_step": ts * 1e3, "edges_per_s": E / ts, "edge_kernel_ms": te * 1e3,
                    "node_side_ms": profs["node_step"]["avg_ms"],
                    "roofline": {"bound": "mfma", "kernel": "k_edge_ring", "achieved": 6 * flops_edge_kernel(E) / te / 1e12 if te > 0 else 0.0,
                                 "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": 6 * flops_edge_kernel(E) / te / 1e12 / PEAK_BF16_MFMA_TFLOPS if te > 0 else 0.0}}
                engs.close()
            finally:
                lib.mgn_debug_split_f16(oldh)
        if world == 1 and not args.no_secondary and args.dtype == "f32":
            engb = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank, dtype="bf16")
            engb.set_params(ps)
            engb.set_graph(s, r, N)
            engb.latents_randn(1234)
            dtb, profb = time_single(engb, 6, 2, barrier_sync)          # (six passes: the scattered-label ratios below divide by it)
            tb = dtb / (6 * MPS)
            bytes_b = 2.0 * L * (2 * E + 2 * N) + 8.0 * E
            out["bf16"] = {"workload": "same M-1M mesh, bf16 storage + bf16 MFMA (BASELINE.json configs[2] precision; single edge set)",
                           "ms_per_processor_step": tb * 1e3, "edges_per_s": E / tb, "edge_kernel_ms": profb["edge_step"]["avg_ms"],
                           "node_side_ms": profb["node_step"]["avg_ms"], "algorithmic_GBps": bytes_b / tb / 1e9,
                           "hbm_frac_of_8TBps": bytes_b / tb / 1e9 / 8000.0,
                           "mfma_TFLOPs_algorithmic": flops_algorithmic(E, N) / tb / 1e12}
            eb_b = 2.0 * L * 2 * E + 8.0 * E + 3.0 * 2.0 * L * N
            te_b = profb["edge_step"]["avg_ms"] * 1e-3
            trb, srcb = committed_traffic(["k_edge_bf16"], "bf16") if args.nx == 1000 else (None, None)
            out["bf16"]["roofline"] = {"bound": "hbm", "kernel": "k_edge_bf16", "achieved": eb_b / te_b / 1e9 if te_b > 0 else 0.0,
                                       "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": eb_b / te_b / 1e9 / PEAK_HBM_GBPS if te_b > 0 else 0.0,
                                       "bytes_per_launch": eb_b, "traffic": trb, "traffic_source": srcb,
                                       "processor_step_frac": bytes_b / tb / 1e9 / PEAK_HBM_GBPS}
            engb.close()
        if world == 1 and not args.no_secondary and args.dtype == "f32":
            # The same mesh under arbitrary node labels (DeepMind's trajectories; create_base_graph passes them through, reference
            # src/graph.jl:30-36): mgn_set_graph re-numbers it breadth-first inside the engine (csrc/graph_host.cpp) -- against the
            # coherent labels above, and against the scattered labels kept as they are (MGN_RENUMBER=0)
            import ctypes as _C
            lib_r = mgn_amd.load()
            lib_r.mgn_debug_renumber.restype = _C.c_int
            lib_r.mgn_debug_renumber.argtypes = [_C.c_int]
            perm = np.random.default_rng(99).permutation(N).astype(np.int32)
            sp, rp = perm[s], perm[r]
            scat = {"workload": "same M-1M mesh, node labels randomly permuted; `renumbered`: the default (breadth-first order inside "
                                "mgn_set_graph), `kept`: MGN_RENUMBER=0"}
            for pol, key in ((1, "renumbered"), (0, "kept")):
                oldp = lib_r.mgn_debug_renumber(pol)
                try:
                    for dtn in ("f32", "bf16"):
                        engp = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank, dtype=dtn)
                        engp.set_params(ps)
                        t0 = time.perf_counter()
                        engp.set_graph(sp, rp, N)
                        tset = time.perf_counter() - t0
                        engp.latents_randn(1234)
                        dtp_, profp = time_single(engp, 6, 2, barrier_sync)
                        scat[f"{key}_{dtn}"] = {"ms_per_processor_step": dtp_ / (6 * MPS) * 1e3, "edge_kernel_ms": profp["edge_step"]["avg_ms"],
                                                "node_side_ms": profp["node_step"]["avg_ms"], "graph_setup_s": tset}
                        engp.close()
                finally:
                    lib_r.mgn_debug_renumber(oldp)
            scat["vs_coherent_f32"] = scat["renumbered_f32"]["ms_per_processor_step"] / (t_step * 1e3)
            scat["vs_coherent_bf16"] = scat["renumbered_bf16"]["ms_per_processor_step"] / out["bf16"]["ms_per_processor_step"]
            out["scattered_labels"] = scat
        if world == 1 and not args.no_secondary and args.dtype == "f32":
            pos2, cells2, _, _ = mgn_amd.synth.mesh_cyl(1234, 2000)
            s2, r2 = mgn_amd.synth.cells_to_edges(cells2)
            eng2 = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank)
            eng2.set_params(ps)
            eng2.set_graph(s2, r2, pos2.shape[0])
            eng2.latents_randn(1234)
            k2 = max(args.steps, 50)
            _, prof2 = time_single(eng2, 5, 3, barrier_sync)          # per-kernel times (event records on)
            # wall time: no events, hipGraph replay.  Best of three passes after a 0.2 s warm-up: on a fresh box the first
            # pass of this latency-bound section has been seen 2.3 x slower than every later one (clocks / first-use effects)
            tw0 = time.perf_counter()
            while time.perf_counter() - tw0 < 0.2:
                eng2.processor_steps_dev(MPS)
            dt2 = None
            for _ in range(3):
                barrier_sync()
                t0 = time.perf_counter()
                for _ in range(k2):
                    eng2.processor_steps_dev(MPS)
                barrier_sync()
                dtp = time.perf_counter() - t0
                dt2 = dtp if dt2 is None else min(dt2, dtp)
            out["secondary"] = {"workload": f"M-cyl Delaunay 2000 pts: N={pos2.shape[0]}, E={s2.size}, L=128, 15 steps, fp32 (BASELINE.json configs[1])",
                                "edges_per_s": s2.size * MPS * k2 / dt2, "nodes_per_s": pos2.shape[0] * MPS * k2 / dt2,
                                "us_per_processor_step": dt2 / (k2 * MPS) * 1e6,
                                "edge_kernel_us": prof2["edge_step"]["avg_ms"] * 1e3, "node_kernel_us": prof2["node_step"]["avg_ms"] * 1e3}
            # cfg-5: 100-step inference rollout on the same mesh through the native driver (mgn_rollout):
            # full Encode-Process-Decode per right-hand side, Euler (100 RHS) and adaptive Tsit5 (6 RHS per step)
            pos5, cells5, ntype5, vel5 = mgn_amd.synth.mesh_cyl(1234, 2000)
            onehot5 = np.eye(7, dtype=np.float32)[ntype5]
            rel5 = pos5[s2] - pos5[r2]
            ef5 = np.concatenate([rel5, np.linalg.norm(rel5, axis=1, keepdims=True)], 1).astype(np.float32)
            eng2.set_norms(node=(np.ones(FN, np.float32), np.zeros(FN, np.float32)),
                           edge=(1.0 / np.maximum(ef5.std(0), 1e-8), -ef5.mean(0) / np.maximum(ef5.std(0), 1e-8)),
                           out=(np.full(O, 0.05, np.float32), np.zeros(O, np.float32)))
            vm5 = np.isin(ntype5, [0, 5]).astype(np.float32)
            roll = {}
            for name, kw in (("Euler", dict(dt=0.01)), ("Tsit5", dict())):
                eng2.rollout(name, vel5, onehot5, ef5, 0.0, 0.1, 0.01, 11, val_mask=vm5, **kw)          # warm-up
                t0 = time.perf_counter()
                _, st5 = eng2.rollout(name, vel5, onehot5, ef5, 0.0, 1.0, 0.01, 101, val_mask=vm5, **kw)
                dt5 = time.perf_counter() - t0
                roll[name] = {"ms_per_rollout": dt5 * 1e3, "rhs_evals": st5["n_rhs"], "accepted_steps": st5["n_accept"],
                              "rejected_steps": st5["n_reject"], "us_per_rhs": dt5 / max(st5["n_rhs"], 1) * 1e6}
            out["secondary"]["rollout_100_saves"] = {"workload": "cfg-5 shaped: M-cyl, t in [0,1], saveat 0:0.01:1, random-init weights "
                                                     "(BASELINE.json configs[4]); native driver, host in/out included", **roll}
            eng2.close()
            # N2: one training step (step!: forward with kept activations, backward, all gradients) on the same datapoint
            engt = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank)
            engt.set_params(ps)
            engt.set_graph(s2, r2, pos2.shape[0])
            rngt = np.random.default_rng(0)
            nft = rngt.standard_normal((pos2.shape[0], FN)).astype(np.float32)
            eft = rngt.standard_normal((s2.size, FE)).astype(np.float32)
            tgt = rngt.standard_normal((pos2.shape[0], O)).astype(np.float32)
            maskt = np.nonzero(np.isin(ntype5, [0, 5]))[0].astype(np.int32)
            for _ in range(3):          # eager, hipGraph capture, first replay
                engt.step(nft, eft, tgt, maskt)
            t0 = time.perf_counter()
            for _ in range(10):
                _, loss_t = engt.step(nft, eft, tgt, maskt)
            dtt = (time.perf_counter() - t0) / 10
            # the reference keeps graph, ps and gs on the GPU: same call with device arrays (no 9 MB gradient download per step)
            dev = lambda a_: torch.from_numpy(a_).to(f"cuda:{local_rank}")
            nft_d, eft_d, tgt_d = dev(nft), dev(eft), dev(tgt)
            gs_d = torch.zeros(engt.param_count, device=f"cuda:{local_rank}")
            engt.step(nft_d, eft_d, tgt_d, maskt, out=gs_d)
            t0 = time.perf_counter()
            for _ in range(10):
                engt.step(nft_d, eft_d, tgt_d, maskt, out=gs_d)
            dtd = (time.perf_counter() - t0) / 10
            # the loop the reference runs (src/MeshGraphNets.jl:375-377): the optimiser changes ps, so every iteration is
            # mgn_set_params + mgn_step -- the parameters go to the device once and the training layouts are packed there
            ps_it = ps.copy()
            for _ in range(3):          # (the device-array calls above used other buffers: the launch graphs are captured again first)
                ps_it *= np.float32(1.00001)
                engt.set_params(ps_it)
                engt.step(nft, eft, tgt, maskt)
            t0 = time.perf_counter()
            for _ in range(10):
                ps_it *= np.float32(1.00001)
                engt.set_params(ps_it)
                engt.step(nft, eft, tgt, maskt)
            dtl = (time.perf_counter() - t0) / 10
            out["secondary"]["train_step"] = {"ms_per_iteration_with_new_params": dtl * 1e3,
                                              "workload": "mgn_step == step! (src/strategies.jl:418-422) on the M-cyl datapoint, L=128, 15 steps, fp32; "
                                                          "host in/out included", "ms_per_step": dtt * 1e3,
                                              "ms_per_step_device_arrays": dtd * 1e3, "loss_finite": bool(np.isfinite(loss_t))}
            engt.close()
            # N2 at the headline size: one training step on M-1M itself (61 GB with every step recomputed + 10.7 GB per step stored)
            try:
                engT = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank)
                engT.set_params(ps)
                engT.set_graph(s, r, N)
                rngT = np.random.default_rng(0)
                nfT = rngT.standard_normal((N, FN), dtype=np.float32)
                efT = rngT.standard_normal((E, FE), dtype=np.float32)
                tgT = rngT.standard_normal((N, O), dtype=np.float32)
                mkT = np.arange(0, N, 2, dtype=np.int32)
                engT.step(nfT, efT, tgT, mkT)                        # first call: weights repacked, arena allocated
                t0 = time.perf_counter()
                _, lossT = engT.step(nfT, efT, tgT, mkT)
                dtT = time.perf_counter() - t0
                import ctypes as _C
                engT.lib.mgn_debug_train_keep_steps.argtypes = [_C.c_void_p]
                keepT = int(engT.lib.mgn_debug_train_keep_steps(engT.h))   # processor steps whose H1 / H2 / Y are stored, not recomputed
                # MFMA flops one step executes (fp32, v_mfma_f32_32x32x2_f32; first edge layer factored as in the forward kernels):
                # per processor step  forward (+ recomputation where the step's activations are not stored) 1 or 2 x (98 304 E + 196 608 N), backward (transposed chunks) 98 304 E + 196 608 N,
                # weight gradients 98 304 E + 196 608 N  (docs/experiments.md, training step)
                flT = (4.0 * MPS - max(keepT, 0)) * (98304.0 * E + 196608.0 * N)
                # forward / recomputation / backward / layer-1 halves and (round 6: k_wgrad_h2) the weight gradients run on two fp16 pieces:
                # 3 piece products per fp32 product on the 16-bit pipe (MGN_WGRAD_H2=0: the weight gradients on the fp32 MFMA pipe)
                wg_h2 = os.environ.get("MGN_WGRAD_H2", "1") != "0"
                fl32 = 0.0 if wg_h2 else MPS * (98304.0 * E + 196608.0 * N)
                fl16 = 3.0 * (flT - fl32)
                floor_s = fl16 / (PEAK_BF16_MFMA_TFLOPS * 1e12) + fl32 / (PEAK_F32_MFMA_TFLOPS * 1e12)
                out["secondary"]["train_step_1m"] = {
                    "workload": "mgn_step == step! on M-1M (N = 1 000 000, E = 5 992 002, L = 128, 15 steps; fp32 storage, forward / recomputation / "
                                "backward MLP chains and weight gradients on two fp16 pieces (MGN_TRAIN_F16=0: fp32 MFMA), the aggregation of e' inside the forward's edge "
                                "launch and the LayerNorm-parameter sums inside the backward launches; "
                                "activations stored for `stored_steps` of the 15 processor steps -- as many as the device's free memory holds -- and "
                                "recomputed in the reverse pass for the others); host in/out included",
                    "s_per_step": dtT, "stored_steps": keepT, "loss_finite": bool(np.isfinite(lossT)),
                    "roofline": {"bound": "mfma", "fp32_products_per_step": flT, "executed_flops_16bit": fl16, "executed_flops_fp32": fl32,
                                 "matrix_floor_s": floor_s, "frac": floor_s / dtT, "fp32_equivalent_TFLOPs": flT / dtT / 1e12,
                                 "unit": "s", "achieved": dtT, "peak": floor_s,
                                 "note": "matrix floor = 16-bit piece products at the dense 16-bit peak (+ fp32 weight-gradient products at 157.3 TFLOP/s "
                                         "under MGN_WGRAD_H2=0); the step is HBM-bound: forward, backward and weight-gradient launches move their "
                                         "20.7 / 30.9 / 15.6 GB at 4.7-5.4 TB/s (profiles/r06); processor "
                                         "MLPs only (encoders / decoder, segmented sums, reductions not counted)"}}
                engT.close()
                # ... and with NO step stored (every processor MLP's forward recomputed in the reverse pass: what a device that is not
                # empty gets; 61 GB instead of 61 + 10.7 per stored step)
                os.environ["MGN_TRAIN_KEEP_STEPS"] = "0"
                try:
                    engT = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank)
                    engT.set_params(ps)
                    engT.set_graph(s, r, N)
                    engT.step(nfT, efT, tgT, mkT)
                    t0 = time.perf_counter()
                    _, lossT0 = engT.step(nfT, efT, tgT, mkT)
                    dtT0 = time.perf_counter() - t0
                    out["secondary"]["train_step_1m"]["s_per_step_no_stored_steps"] = dtT0
                    out["secondary"]["train_step_1m"]["same_loss_without_stored_steps"] = bool(lossT0 == lossT)
                    engT.close()
                finally:
                    del os.environ["MGN_TRAIN_KEEP_STEPS"]
                del nfT, efT, tgT
            except Exception as ex:   # noqa: BLE001  (a box with less free memory than the 61 GB this needs at least)
                out["secondary"]["train_step_1m"] = {"error": str(ex)[:200]}
            # spec hedge: whole-array LayerNorm (mgn_config.ln_dims = MGN_LN_ALL) -- the unfused driver's cost, from whole
            # forwards (small host arrays in and out): one forward = encoders + 15 processor steps + decoder on the device
            try:
                engA = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank, ln_dims="all")
                engA.set_params(ps)
                engA.set_graph(s, r, N)
                rngA = np.random.default_rng(1)
                nfA = rngA.standard_normal((N, FN), dtype=np.float32)
                efA = rngA.standard_normal((E, FE), dtype=np.float32)
                engA.forward(nfA, efA)
                t0 = time.perf_counter()
                outA = engA.forward(nfA, efA)
                dtA = time.perf_counter() - t0
                out["secondary"]["whole_array_layernorm_1m"] = {
                    "workload": "mgn_forward on M-1M under ln_dims = MGN_LN_ALL (what Lux 0.5's LayerNorm computes at dims = Colon()): per MLP the "
                                "streaming MLP kernel without LayerNorm, grid-wide statistics in double, an apply pass; host in/out (132 MB) included",
                    "ms_per_forward": dtA * 1e3, "ms_per_processor_step_upper_bound": dtA * 1e3 / MPS, "finite": bool(np.isfinite(outA).all())}
                engA.close()
                del nfA, efA
            except Exception as ex:   # noqa: BLE001
                out["secondary"]["whole_array_layernorm_1m"] = {"error": str(ex)[:200]}
            # the same hedge on the cylinder mesh: right-hand side (resident inputs), 100-save Euler rollout, training step
            try:
                engB = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank, ln_dims="all")
                engB.set_params(ps)
                engB.set_graph(s2, r2, pos2.shape[0])
                engB.set_static(onehot5, ef5, vm5)
                for _ in range(4):
                    engB.ode_step(vel5)
                t0 = time.perf_counter()
                for _ in range(20):
                    engB.ode_step(vel5)
                rhsB = (time.perf_counter() - t0) / 20
                engB.rollout("Euler", vel5, onehot5, ef5, 0.0, 0.1, 0.01, 11, dt=0.01, val_mask=vm5)
                t0 = time.perf_counter()
                engB.rollout("Euler", vel5, onehot5, ef5, 0.0, 1.0, 0.01, 101, dt=0.01, val_mask=vm5)
                roB = time.perf_counter() - t0
                for _ in range(3):
                    engB.step(nft, eft, tgt, maskt)
                t0 = time.perf_counter()
                for _ in range(10):
                    engB.step(nft, eft, tgt, maskt)
                stB = (time.perf_counter() - t0) / 10
                out["secondary"]["whole_array_layernorm_cyl"] = {
                    "workload": "M-cyl under ln_dims = MGN_LN_ALL: mgn_ode_step on resident inputs (launch-graph replay), native Euler rollout with 100 saves, "
                                "mgn_step; host in/out included",
                    "us_per_rhs": rhsB * 1e6, "euler_100_saves_ms": roB * 1e3, "train_step_ms": stB * 1e3}
                engB.close()
            except Exception as ex:   # noqa: BLE001
                out["secondary"]["whole_array_layernorm_cyl"] = {"error": str(ex)[:200]}
            # mid-size meshes (real CFD meshes, and the per-GPU share of M-1M on 8 GPUs): where the kernel families meet
            mids = {}
            for nxm in (128, 300, 354):   # 354 x 354 = the per-GPU share of M-1M on 8 GPUs
                posm, sm, rm = mgn_amd.synth.mesh_1m(1234, nxm, nxm)
                engm = mgn_amd.Engine(FN, FE, O, L, 2, MPS, device=local_rank)
                engm.set_params(ps)
                engm.set_graph(sm, rm, posm.shape[0])
                engm.latents_randn(1234)
                for _ in range(3):
                    engm.processor_steps_dev(MPS)
                barrier_sync()
                km = 10
                t0 = time.perf_counter()
                for _ in range(km):
                    engm.processor_steps_dev(MPS)
                barrier_sync()
                dtm = time.perf_counter() - t0
                mids[f"N={posm.shape[0]},E={sm.size}"] = {"us_per_processor_step": dtm / (km * MPS) * 1e6, "edges_per_s": sm.size * MPS * km / dtm}
                engm.close()
            out["secondary"]["mid_size_meshes"] = mids
            # the shapes beside the headline's (hidden_layers and layer_size are free integers, a second edge set exists): what a
            # processor step costs on the 125 k-node share there -- the fp32-MFMA generic kernels (hidden_layers != 2, L != 128) and the
            # two-edge-set node side (three bf16 pieces) have not been moved to the streamed fp16 kernels; random small parameters, timing only
            try:
                shapes = {}
                posm, sm, rm = mgn_amd.synth.mesh_1m(1234, 354, 354)
                Nm = posm.shape[0]

                def time_steps(engx, nedges):
                    engx.latents_randn(1234)
                    for _ in range(3):
                        engx.processor_steps_dev(MPS)
                    barrier_sync()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        engx.processor_steps_dev(MPS)
                    barrier_sync()
                    dtx = time.perf_counter() - t0
                    v_, e_ = engx.latents_export()
                    return {"us_per_processor_step": dtx / (10 * MPS) * 1e6, "edges_per_s": nedges * MPS * 10 / dtx, "finite": bool(np.isfinite(v_).all() and np.isfinite(e_).all())}

                for name, (Lx, hl) in {"L=128,hidden_layers=3": (128, 3), "L=128,hidden_layers=1": (128, 1), "L=64,hidden_layers=2": (64, 2)}.items():
                    engx = mgn_amd.Engine(FN, FE, O, Lx, hl, MPS, device=local_rank)
                    engx.set_params((np.random.default_rng(5).standard_normal(engx.param_count) * 0.08).astype(np.float32))
                    engx.set_graph(sm, rm, Nm)
                    shapes[name] = time_steps(engx, sm.size)
                    engx.close()
                mf = mgn_amd.synth.mesh_flag(nx=354, ny=354, radius=0.0035)
                eng2 = mgn_amd.Engine(12, 7, 3, L, 2, MPS, device=local_rank, Fe2=4)
                eng2.set_params((np.random.default_rng(6).standard_normal(eng2.param_count) * 0.08).astype(np.float32))
                eng2.set_graph(mf["s"], mf["r"], mf["mesh_pos"].shape[0])
                eng2.set_edge_set(1, mf["s2"], mf["r2"])
                shapes[f"L=128,two edge sets (mesh E={mf['s'].size}, world E={mf['s2'].size})"] = time_steps(eng2, mf["s"].size + mf["s2"].size)
                eng2.close()
                shapes["same mesh, default shape"] = mids.get(f"N={Nm},E={sm.size}")
                out["secondary"]["other_shapes_125k"] = shapes
            except Exception as ex:   # noqa: BLE001
                out["secondary"]["other_shapes_125k"] = {"error": str(ex)[:300]}
            # what the memory system gives a plain device copy of an array of the e latents' size (read + write counted): the yardstick for
            # the bytes per second the processor kernels move by the counters (roofline.traffic / avg_launch_ms)
            try:
                nbytes = int(E) * L * 4
                src_t = torch.empty(nbytes // 4, dtype=torch.float32, device=f"cuda:{local_rank}").normal_()
                dst_t = torch.empty_like(src_t)
                for _ in range(3):
                    dst_t.copy_(src_t)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    dst_t.copy_(src_t)
                torch.cuda.synchronize()
                tc = (time.perf_counter() - t0) / 10
                out["secondary"]["memory_system"] = {"device_copy_GBps_read_plus_write": 2.0 * nbytes / tc / 1e9, "bytes_copied": nbytes,
                                                     "note": "torch tensor copy (a streaming kernel) of an array of the e latents' size on this box; the edge "
                                                             "kernel moves `roofline.traffic` bytes per launch in `avg_launch_ms`"}
                del src_t, dst_t
            except Exception as ex:   # noqa: BLE001
                out["secondary"]["memory_system"] = {"error": str(ex)[:200]}
            try:
                out["secondary"]["scaling_model"] = scaling_model(ps, pos, s, r, N, E, t_step, local_rank, barrier_sync)
            except Exception as ex:   # noqa: BLE001
                out["secondary"]["scaling_model"] = {"error": str(ex)[:300]}
            # cfg-3: flag_simple-shaped cloth, mesh + world edges (two edge sets), 15 steps, bf16 (and fp32 beside it)
            mf = mgn_amd.synth.mesh_flag()
            Nf, Ef, Ef2 = mf["mesh_pos"].shape[0], mf["s"].size, mf["s2"].size
            psf = glorot_params(1234, 12, 7, 3, Fe2=4)
            flag = {"workload": f"M-flag 40x40 folded cloth: N={Nf}, mesh E={Ef} (Fe=7), world E={Ef2} (Fe=4), L=128, 15 steps "
                                "(BASELINE.json configs[2]); hipGraph replay, latents resident"}
            for dt_name in ("bf16", "f32"):
                eng3 = mgn_amd.Engine(12, 7, 3, L, 2, MPS, device=local_rank, dtype=dt_name, Fe2=4)
                eng3.set_params(psf)
                eng3.set_graph(mf["s"], mf["r"], Nf)
                eng3.set_edge_set(1, mf["s2"], mf["r2"])
                eng3.latents_randn(1234)
                for _ in range(5):
                    eng3.processor_steps_dev(MPS)
                barrier_sync()
                k3 = max(args.steps, 50)
                t0 = time.perf_counter()
                for _ in range(k3):
                    eng3.processor_steps_dev(MPS)
                barrier_sync()
                dt3 = time.perf_counter() - t0
                flag[dt_name] = {"us_per_processor_step": dt3 / (k3 * MPS) * 1e6, "edges_per_s": (Ef + Ef2) * MPS * k3 / dt3,
                                 "nodes_per_s": Nf * MPS * k3 / dt3}
                eng3.close()
            out["secondary"]["flag_two_edge_sets"] = flag
            # N3: the graph prologue (create_base_graph, src/graph.jl:25-55) at the M-1M size, host loops vs device kernels
            _, cells1m = mgn_amd.synth.grid_mesh(args.nx, args.nx, 1234)
            engp = mgn_amd.Engine(12, 7, 3, L, 2, MPS, device=local_rank, Fe2=4)
            t0 = time.perf_counter()
            sh, rh = mgn_amd.triangles_to_edges_native(cells1m)
            t_th = time.perf_counter() - t0
            engp.triangles_to_edges_dev(cells1m[:1024])
            t0 = time.perf_counter()
            sd, rd = engp.triangles_to_edges_dev(cells1m)
            t_td = time.perf_counter() - t0
            big = np.random.default_rng(0).random((200000, 3)).astype(np.float32)
            none = np.zeros(0, np.int32)
            t0 = time.perf_counter()
            swh, rwh = mgn_amd.world_edges_native(big, 0.012, none, none)
            t_wh = time.perf_counter() - t0
            engp.set_graph(none, none, 200000)
            engp.world_edges_dev(1, big, 0.012)
            t0 = time.perf_counter()
            n_w = engp.world_edges_dev(1, big, 0.012)
            t_wd = time.perf_counter() - t0
            out["secondary"]["graph_prologue"] = {
                "triangles_to_edges": {"cells": int(cells1m.shape[0]), "directed_edges": int(sd.size), "host_s": t_th, "device_s": t_td,
                                       "identical": bool(np.array_equal(sd, sh) and np.array_equal(rd, rh)),
                                       "note": "device: radix sort / unique of packed 64-bit keys, PCIe in and out included"},
                "world_edges_200k_nodes": {"edges": int(n_w), "host_search_s": t_wh, "device_search_and_install_s": t_wd,
                                           "identical_count": bool(n_w == swh.size)},
                "set_graph_s": t_setup}
            engp.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ps)
        print(json.dumps(out), flush=True)

    if staged:
        eng.comm_barrier()
    eng.close()
    del store


if __name__ == "__main__":
    main()
