"""Host-side mirror of the GraphNetCore surface the reference calls, over the C ABI.

Names follow the reference so that the parity tests read like its call sites:
    FeatureGraph(nf, ef, senders, receivers)              reference src/graph.jl:87-96
    GraphNetwork{model, ps, st, e_norm, n_norm, o_norm}   fields used at src/solve.jl:200-208, src/graph.jl:80-93
    mgn.model(graph, ps, st) -> (output, st)              src/solve.jl:200

Array convention: NumPy C row-major [count][feat] == the bytes of Julia's (feat x count) arrays.
There is no CPU compute path: constructing an Engine without the built HIP extension or without a
GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import MgnConfig, f32, i32, i64


class MgnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_capi.STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


def _c32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"DimensionMismatch: expected {tuple(shape)}, got {tuple(a.shape)}")
    return a


def _host_or_device(a, shape, writable=False):
    """(object kept alive, float* for the C ABI) of a NumPy array or a contiguous fp32 torch tensor (host or device: the
    entry points that document it copy with hipMemcpyDefault)."""
    if hasattr(a, "data_ptr"):                       # torch tensor
        import torch
        if a.dtype != torch.float32 or not a.is_contiguous():
            raise ValueError("tensor arguments must be contiguous float32")
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"DimensionMismatch: expected {tuple(shape)}, got {tuple(a.shape)}")
        return a, C.cast(C.c_void_p(a.data_ptr()), C.POINTER(C.c_float))
    if writable:
        if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous and a.flags.writeable):
            raise ValueError("out must be a writable contiguous float32 array")
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"DimensionMismatch: expected {tuple(shape)}, got {tuple(a.shape)}")
        return a, f32(a)
    a = _c32(a, shape)
    return a, f32(a)


class Engine:
    """One engine handle == one mesh partition on one GPU (rank/nranks select the partition)."""

    def __init__(self, Fn, Fe, O, L=128, hidden_layers=2, mps=15, rank=0, nranks=1, device=-1, dtype="f32", Fe2=None, ln_mode=0, ln_dims=0):
        """Fe2: input width of a second edge set (world edges, MGN-spec "per edge set"); None = the reference's one set.
        ln_mode: 0 = (x - mean) / sqrt(var + eps) (MGN-spec v1), 1 = (x - mean) / (sqrt(var) + eps) (spec_variant)."""
        self.lib = _capi.load()
        self.cfg = MgnConfig(Fn, Fe, O, L, hidden_layers, mps, {"f32": 0, "bf16": 1}[dtype], rank, nranks, device,
                             2 if Fe2 else 1, Fe2 or 0, ln_mode, {0: 0, 1: 1, "rows": 0, "all": 1}[ln_dims])
        self.E2 = 0
        self.h = C.c_void_p()
        rc = self.lib.mgn_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise MgnError(rc, self.lib.mgn_last_error(None).decode())
        self.N = self.E = 0
        self.n_own = self.n_halo = self.e_local = 0

    # -- lifecycle -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mgn_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MgnError(rc, self.lib.mgn_last_error(self.h).decode())

    @property
    def host_only(self):
        return self.cfg.device == _capi.MGN_DEVICE_NONE

    def set_stream(self, stream_ptr):
        """stream_ptr: raw hipStream_t (0 = HIP default stream, e.g. torch.cuda.current_stream().cuda_stream);
        None = the engine's own private stream (MGN_STREAM_OWN)."""
        ptr = C.c_void_p(-1) if stream_ptr is None else C.c_void_p(stream_ptr)
        self._chk(self.lib.mgn_set_stream(self.h, ptr))

    def synchronize(self):
        self._chk(self.lib.mgn_synchronize(self.h))

    # -- parameters ----------------------------------------------------------------------------
    @property
    def param_count(self):
        return int(self.lib.mgn_param_count(C.byref(self.cfg)))

    def set_params(self, packed):
        packed = _c32(packed).ravel()
        self._chk(self.lib.mgn_set_params(self.h, f32(packed), packed.size))

    def get_params(self):
        out = np.empty(self.param_count, np.float32)
        self._chk(self.lib.mgn_get_params(self.h, f32(out), out.size))
        return out

    def set_norms(self, node=None, edge=None, out=None):
        """Each argument: None (identity) or (scale[F], shift[F]) with forward y = x*scale + shift;
        `out` is the INVERSE map of o_norm (inverse_data)."""
        def sp(p, n):
            if p is None:
                return None, None
            return _c32(p[0], (n,)), _c32(p[1], (n,))
        ns, nsh = sp(node, self.cfg.Fn)
        es, esh = sp(edge, self.cfg.Fe)
        os_, osh = sp(out, self.cfg.O)
        self._keep_norms = (ns, nsh, es, esh, os_, osh)
        self._chk(self.lib.mgn_set_norms(self.h, f32(ns), f32(nsh), f32(es), f32(esh), f32(os_), f32(osh)))

    # -- graph ---------------------------------------------------------------------------------
    def set_graph(self, senders, receivers, N, index_base=0, mesh_pos=None):
        s = np.ascontiguousarray(senders, dtype=np.int32).ravel()
        r = np.ascontiguousarray(receivers, dtype=np.int32).ravel()
        if s.size != r.size:
            raise ValueError("DimensionMismatch: senders and receivers differ in length")
        pos, pd = None, 0
        if mesh_pos is not None:
            pos = _c32(mesh_pos)
            if pos.ndim != 2 or pos.shape[0] != N:
                raise ValueError("DimensionMismatch: mesh_pos must be [N][dim]")
            pd = pos.shape[1]
        self._chk(self.lib.mgn_set_graph(self.h, N, s.size, i32(s), i32(r), index_base, f32(pos), pd))
        self.N, self.E, self.E2 = int(N), int(s.size), 0
        self._refresh_partition()

    @staticmethod
    def partition_nodes(N, nranks, mesh_pos=None):
        """owner[N]: the node partition mgn_set_graph derives (recursive coordinate bisection of mesh_pos, or index blocks)."""
        lib = _capi.load()
        owner = np.empty(int(N), np.int32)
        pos, pd = None, 0
        if mesh_pos is not None:
            pos = _c32(mesh_pos)
            pd = pos.shape[1]
        rc = lib.mgn_partition_nodes(int(N), f32(pos), pd, int(nranks), i32(owner))
        if rc != 0:
            raise MgnError(rc, "mgn_partition_nodes")
        return owner

    def set_graph_local(self, senders, receivers, N, owner, index_base=0):
        """Rank-local ingest: the same state as set_graph(senders, receivers, N) with the partition `owner`, from the edges this rank
        has an end of (filtered here with numpy; a caller that holds only its part passes it to mgn_set_graph_local directly)."""
        s = np.ascontiguousarray(senders, dtype=np.int32).ravel()
        r = np.ascontiguousarray(receivers, dtype=np.int32).ravel()
        owner = np.ascontiguousarray(owner, dtype=np.int32).ravel()
        rank = self.cfg.rank
        touch = np.nonzero((owner[s - index_base] == rank) | (owner[r - index_base] == rank))[0].astype(np.int64)
        st, rt = np.ascontiguousarray(s[touch]), np.ascontiguousarray(r[touch])
        self._chk(self.lib.mgn_set_graph_local(self.h, int(N), i32(owner), int(s.size), int(touch.size), i32(st), i32(rt), i64(touch), index_base))
        self.N, self.E, self.E2 = int(N), int(s.size), 0
        self._refresh_partition()

    def _refresh_partition(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int64()
        self._chk(self.lib.mgn_partition_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        self.n_own, self.n_halo, self.e_local = a.value, b.value, c.value

    def set_edge_set(self, set_index, senders, receivers, index_base=0):
        """Topology of the second edge set (after set_graph; as often as it changes)."""
        s = np.ascontiguousarray(senders, dtype=np.int32).ravel()
        r = np.ascontiguousarray(receivers, dtype=np.int32).ravel()
        if s.size != r.size:
            raise ValueError("DimensionMismatch: senders and receivers differ in length")
        self._chk(self.lib.mgn_set_edge_set(self.h, set_index, s.size, i32(s), i32(r), index_base))
        self.E2 = int(s.size)
        self._refresh_partition()

    def set_edge_features(self, set_index, ef):
        ef = _c32(ef, (self.E2, self.cfg.Fe2))
        self._chk(self.lib.mgn_set_edge_features(self.h, set_index, f32(ef)))

    def edge_set_info(self, set_index):
        a, b = C.c_int64(), C.c_int64()
        self._chk(self.lib.mgn_edge_set_info(self.h, set_index, C.byref(a), C.byref(b)))
        return a.value, b.value

    def edge_latents_import(self, set_index, e):
        e = _c32(e, ((self.E, self.E2)[set_index], self.cfg.L))
        self._chk(self.lib.mgn_edge_latents_import(self.h, set_index, f32(e)))

    def edge_latents_export(self, set_index, e=None):
        e = np.zeros(((self.E, self.E2)[set_index], self.cfg.L), np.float32) if e is None else e
        self._chk(self.lib.mgn_edge_latents_export(self.h, set_index, f32(e)))
        return e

    def owned_nodes(self):
        out = np.empty(self.n_own, np.int32)
        self._chk(self.lib.mgn_owned_nodes(self.h, i32(out)))
        return out

    def local_edges(self):
        out = np.empty(self.e_local, np.int64)
        self._chk(self.lib.mgn_local_edges(self.h, i64(out)))
        return out

    def halo_counts(self):
        n = self.cfg.nranks
        s, r = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self._chk(self.lib.mgn_halo_counts(self.h, i32(s), i32(r)))
        return s, r

    def halo_nodes(self):
        out = np.empty(self.n_halo, np.int32)
        self._chk(self.lib.mgn_halo_nodes(self.h, i32(out)))
        return out

    def halo_send_index(self):
        s, _ = self.halo_counts()
        out = np.empty(int(s.sum()), np.int32)
        self._chk(self.lib.mgn_halo_send_index(self.h, i32(out)))
        return out

    def local_graph(self):
        snd = np.empty(self.e_local, np.int32)
        rcv = np.empty(self.e_local, np.int32)
        rowptr = np.empty(self.n_own + 1, np.int32)
        self._chk(self.lib.mgn_local_graph(self.h, i32(snd), i32(rcv), i32(rowptr)))
        return snd, rcv, rowptr

    def boundary_count(self):
        n = C.c_int32()
        self._chk(self.lib.mgn_boundary_count(self.h, C.byref(n)))
        return n.value

    def node_owner(self):
        out = np.empty(self.N, np.int32)
        self._chk(self.lib.mgn_node_owner(self.h, i32(out)))
        return out

    # -- model ---------------------------------------------------------------------------------
    def forward(self, nf, ef):
        nf = _c32(nf, (self.N, self.cfg.Fn))
        ef = _c32(ef, (self.E, self.cfg.Fe))
        out = np.zeros((self.N, self.cfg.O), np.float32)
        self._chk(self.lib.mgn_forward(self.h, f32(nf), f32(ef), f32(out)))
        return out

    def set_static(self, node_type_onehot, ef_raw, val_mask=None):
        """Once per trajectory: static RHS inputs become device-resident and the edge encoder runs once; afterwards
        ode_step(x) only moves the state."""
        O, Fn = self.cfg.O, self.cfg.Fn
        oh = _c32(node_type_onehot, (self.N, Fn - O)) if Fn > O else None
        ef = _c32(ef_raw, (self.E, self.cfg.Fe))
        vm = _c32(val_mask, (self.N,)) if val_mask is not None else None
        self._chk(self.lib.mgn_set_static(self.h, f32(oh), f32(ef), f32(vm)))

    # -- graph prologue on the device ----------------------------------------------
    def triangles_to_edges_dev(self, cells):
        """GraphNetCore.triangles_to_edges (reference src/graph.jl:30) on the device: (senders, receivers), two-way, first-occurrence
        order -- the same bits as triangles_to_edges_native."""
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        if cells.ndim != 2 or cells.shape[1] != 3:
            raise ValueError("DimensionMismatch: cells must be [C][3]")
        cap = 6 * cells.shape[0]
        s, r = np.empty(max(cap, 1), np.int32), np.empty(max(cap, 1), np.int32)
        n = C.c_int64()
        self._chk(self.lib.mgn_triangles_to_edges_dev(self.h, cells.ctypes.data, cells.shape[0], s.ctypes.data, r.ctypes.data, cap, C.byref(n)))
        return s[: n.value].copy(), r[: n.value].copy()

    def set_static_mesh(self, node_type, type_min, type_max, mesh_pos, val_mask=None):
        """create_base_graph's feature half on the device (one-hot node types, edge features) + the once-per-trajectory edge encoder."""
        nt = np.ascontiguousarray(node_type, dtype=np.int32).ravel()
        pos = _c32(mesh_pos)
        if nt.size != self.N or pos.shape[0] != self.N:
            raise ValueError("DimensionMismatch: node_type / mesh_pos must have N rows")
        vm = _c32(val_mask, (self.N,)) if val_mask is not None else None
        self._keep_static = (nt, pos, vm)
        self._chk(self.lib.mgn_set_static_mesh(self.h, nt.ctypes.data, type_min, type_max, pos.ctypes.data, pos.shape[1],
                                               vm.ctypes.data if vm is not None else None))

    def world_edges_dev(self, set_index, world_pos, radius):
        """Search the world-edge set on the device and install it (no host round trip).  Returns the number of edges."""
        wp = _c32(world_pos)
        if wp.ndim != 2 or wp.shape[0] != self.N:
            raise ValueError("DimensionMismatch: world_pos must be [N][dim]")
        n = C.c_int64()
        self._chk(self.lib.mgn_world_edges_dev(self.h, set_index, wp.ctypes.data, wp.shape[1], C.c_float(radius), C.byref(n)))
        self.E2 = int(n.value)
        return self.E2

    def edge_set_export(self, set_index):
        E, _ = self.edge_set_info(set_index)
        s, r = np.empty(E, np.int32), np.empty(E, np.int32)
        self._chk(self.lib.mgn_edge_set_export(self.h, set_index, i32(s), i32(r)))
        return s, r

    def ode_step(self, x, node_type_onehot=None, ef_raw=None, val_mask=None):
        O, Fn = self.cfg.O, self.cfg.Fn
        x = _c32(x, (self.N, O))
        if node_type_onehot is None and ef_raw is None and val_mask is None:
            out = np.zeros((self.N, O), np.float32)
            self._chk(self.lib.mgn_ode_step(self.h, f32(x), None, None, None, f32(out)))
            return out
        oh = _c32(node_type_onehot, (self.N, Fn - O)) if Fn > O else None
        ef = _c32(ef_raw, (self.E, self.cfg.Fe))
        vm = _c32(val_mask, (self.N,)) if val_mask is not None else None
        out = np.zeros((self.N, O), np.float32)
        self._chk(self.lib.mgn_ode_step(self.h, f32(x), f32(oh), f32(ef), f32(vm), f32(out)))
        return out

    def rollout(self, solver, x0, node_type_onehot, ef_raw, t0, t1, saves_dt, n_saves, dt=0.0, val_mask=None,
                inflow_mask=None, inflow_data=None, abstol=1e-6, reltol=1e-3, inflow_rule="reference", time_type=np.float32):
        """Native device-side `rollout` (reference src/solve.jl:42-68).  solver: "Euler" (fixed dt) or "Tsit5".
        inflow_rule: "reference" = `floor(Int, t / saves_dt) + 1` in the solver's time type, no tolerance, out of range raises
        (src/solve.jl:151); "tolerant" = + 1e-3, clamped (step k reads frame k).  time_type: np.float32 (the example's `0.0f0:0.01f0:5.99f0`) or np.float64.
        Returns (sol_u [n_saves][N][O], stats dict)."""
        O, Fn = self.cfg.O, self.cfg.Fn
        d = _capi.MgnRolloutDesc()
        d.solver = {"Euler": 0, "Tsit5": 1}[solver]
        d.t0, d.t1, d.dt, d.saves_dt, d.n_saves, d.abstol, d.reltol = t0, t1, dt, saves_dt, n_saves, abstol, reltol
        d.inflow_rule = {"reference": 0, "tolerant": 1}[inflow_rule]
        d.time_f64 = 1 if np.dtype(time_type) == np.float64 else 0
        d.t0_f64, d.t1_f64, d.dt_f64, d.saves_dt_f64 = t0, t1, dt, saves_dt
        x0 = _c32(x0, (self.N, O))
        oh = _c32(node_type_onehot, (self.N, Fn - O)) if Fn > O else None
        ef = _c32(ef_raw, (self.E, self.cfg.Fe))
        vm = _c32(val_mask, (self.N,)) if val_mask is not None else None
        im = np.ascontiguousarray(inflow_mask, dtype=np.uint8).reshape(self.N) if inflow_mask is not None else None
        idata = _c32(inflow_data) if inflow_data is not None else None
        if idata is not None and idata.shape[1:] != (self.N, O):
            raise ValueError("DimensionMismatch: inflow_data must be [frames][N][O]")
        out = np.zeros((n_saves, self.N, O), np.float32)
        d.x0, d.node_type_onehot, d.ef_raw, d.val_mask = f32(x0), f32(oh), f32(ef), f32(vm)
        d.inflow_mask = im.ctypes.data_as(C.POINTER(C.c_uint8)) if im is not None else None
        d.inflow_data = f32(idata)
        d.n_frames = idata.shape[0] if idata is not None else 0
        d.out = f32(out)
        self._chk(self.lib.mgn_rollout(self.h, C.byref(d)))
        return out, dict(n_accept=d.n_accept, n_reject=d.n_reject, n_rhs=d.n_rhs)

    def step(self, nf, ef, target, mask, mask_index_base=0, out=None):
        """step!(mgn, graph, target, mask, mse_reduce) (reference src/strategies.jl:418-422): returns (gs, loss) with gs
        in the packed order of set_params.  nf / ef / target may be NumPy arrays or contiguous fp32 torch tensors on
        the engine's GPU (the reference keeps the graph on the device); `out`: a NumPy array or device tensor of
        param_count floats that receives the gradients (device: no PCIe transfer, the optimiser runs where they are)."""
        nf, p_nf = _host_or_device(nf, (self.N, self.cfg.Fn))
        ef, p_ef = _host_or_device(ef, (self.E, self.cfg.Fe))
        target, p_t = _host_or_device(target, (self.N, self.cfg.O))
        mask = np.ascontiguousarray(mask, dtype=np.int32).ravel()
        if out is None:
            out = np.zeros(self.param_count, np.float32)
        gs, p_gs = _host_or_device(out, (self.param_count,), writable=True)
        loss = C.c_float()
        self._chk(self.lib.mgn_step(self.h, p_nf, p_ef, p_t, i32(mask), mask.size, mask_index_base, p_gs, self.param_count,
                                    C.byref(loss)))
        return gs, loss.value

    def feature_stats(self, x):
        """Per-feature (sum, sum of squares) of x [rows][dim] in float64 on the device: one NormaliserOnline accumulation
        (GraphNetCore; normalisers of reference src/MeshGraphNets.jl:92,193-199).  x: NumPy array or device tensor."""
        shape = tuple(x.shape)
        if len(shape) != 2:
            raise ValueError("DimensionMismatch: x must be [rows][dim]")
        x, px = _host_or_device(x, shape)
        s = np.zeros(shape[1], np.float64)
        q = np.zeros(shape[1], np.float64)
        dp = C.POINTER(C.c_double)
        self._chk(self.lib.mgn_feature_stats(self.h, px, shape[0], shape[1], s.ctypes.data_as(dp), q.ctypes.data_as(dp)))
        return s, q

    def ode_vjp(self, x, node_type_onehot, ef_raw, lam, val_mask=None, want_dxdt=False):
        """lambda^T df/dx and lambda^T df/dps of the RHS f = ode_step (solver-based training, src/strategies.jl:175-196).
        Returns (xbar [N][O], gs [packed], dxdt or None)."""
        O, Fn = self.cfg.O, self.cfg.Fn
        x = _c32(x, (self.N, O))
        lam = _c32(lam, (self.N, O))
        oh = _c32(node_type_onehot, (self.N, Fn - O)) if Fn > O else None
        ef = _c32(ef_raw, (self.E, self.cfg.Fe))
        vm = _c32(val_mask, (self.N,)) if val_mask is not None else None
        xbar = np.zeros((self.N, O), np.float32)
        gs = np.zeros(self.param_count, np.float32)
        dxdt = np.zeros((self.N, O), np.float32) if want_dxdt else None
        self._chk(self.lib.mgn_ode_vjp(self.h, f32(x), f32(oh), f32(ef), f32(vm), f32(lam), f32(dxdt), f32(xbar), f32(gs), gs.size))
        return xbar, gs, dxdt

    def forward_vjp(self, nf, ef, ybar, want_out=False):
        """Pullback of forward() == of `mgn.model(graph, ps, st)` (src/solve.jl:200): ybar^T d out / d nf and ybar^T d out / d ps.
        Returns (nfbar [N][Fn], gs [packed], out or None): what a ChainRulesCore.rrule of the Julia shim's model hands to Zygote."""
        nf = _c32(nf, (self.N, self.cfg.Fn))
        ef = _c32(ef, (self.E, self.cfg.Fe))
        ybar = _c32(ybar, (self.N, self.cfg.O))
        nfbar = np.zeros((self.N, self.cfg.Fn), np.float32)
        gs = np.zeros(self.param_count, np.float32)
        out = np.zeros((self.N, self.cfg.O), np.float32) if want_out else None
        self._chk(self.lib.mgn_forward_vjp(self.h, f32(nf), f32(ef), f32(ybar), f32(out), f32(nfbar), f32(gs), gs.size))
        return nfbar, gs, out

    def processor_steps(self, v, e, nsteps):
        v = _c32(v, (self.N, self.cfg.L)).copy()
        e = _c32(e, (self.E, self.cfg.L)).copy()
        self._chk(self.lib.mgn_processor_steps(self.h, f32(v), f32(e), nsteps))
        return v, e

    # -- device-resident latents ---------------------------------------------------------------
    def latents_import(self, v, e):
        v = _c32(v, (self.N, self.cfg.L))
        e = _c32(e, (self.E, self.cfg.L))
        self._chk(self.lib.mgn_latents_import(self.h, f32(v), f32(e)))

    def latents_export(self, v=None, e=None):
        """Writes owned rows into GLOBAL-shaped arrays (allocated zero-filled when not given)."""
        v = np.zeros((self.N, self.cfg.L), np.float32) if v is None else v
        e = np.zeros((self.E, self.cfg.L), np.float32) if e is None else e
        self._chk(self.lib.mgn_latents_export(self.h, f32(v), f32(e)))
        return v, e

    def latents_randn(self, seed):
        self._chk(self.lib.mgn_latents_randn(self.h, C.c_uint64(seed)))

    def latents_checksum(self):
        a, b, c, d = (C.c_double() for _ in range(4))
        self._chk(self.lib.mgn_latents_checksum(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(sum_v=a.value, sum_e=b.value, sumsq_v=c.value, sumsq_e=d.value)

    def processor_steps_dev(self, nsteps):
        self._chk(self.lib.mgn_processor_steps_dev(self.h, nsteps))

    # -- staged pipeline (multi-partition driver) ------------------------------------------------
    def fwd_upload(self, nf, ef):
        nf = _c32(nf, (self.N, self.cfg.Fn))
        ef = _c32(ef, (self.E, self.cfg.Fe))
        self._keep_in = (nf, ef)
        self._chk(self.lib.mgn_fwd_upload(self.h, f32(nf), f32(ef)))

    def fwd_encode(self):
        self._chk(self.lib.mgn_fwd_encode(self.h))

    def proc_begin(self):
        self._chk(self.lib.mgn_proc_begin(self.h))

    def proc_edge(self, k):
        self._chk(self.lib.mgn_proc_edge(self.h, k))

    def proc_node(self, k, project_next):
        self._chk(self.lib.mgn_proc_node(self.h, k, 1 if project_next else 0))

    def proc_node_phase(self, k, phase):
        """phase 1: node MLP of step k (k = -1: none) + projection of the boundary tiles; phase 2: interior tiles."""
        self._chk(self.lib.mgn_proc_node_phase(self.h, k, phase))

    def proc_edge_phase(self, k, phase):
        """phase 1: edge tiles without halo senders (may run while the halo exchange is in flight); phase 2: the rest."""
        self._chk(self.lib.mgn_proc_edge_phase(self.h, k, phase))

    def edge_boundary_tiles(self, set_index=0):
        a, b = C.c_int32(), C.c_int32()
        self._chk(self.lib.mgn_edge_boundary_tiles(self.h, set_index, C.byref(a), C.byref(b)))
        return a.value, b.value

    def fwd_decode(self):
        self._chk(self.lib.mgn_fwd_decode(self.h))

    def fwd_download(self, out=None):
        out = np.zeros((self.N, self.cfg.O), np.float32) if out is None else out
        self._chk(self.lib.mgn_fwd_download(self.h, f32(out)))
        return out

    @property
    def halo_row_floats(self):
        """width of one halo row in 4-byte units (exchange buffers are float32 tensors; bf16 rows are 64 wide)"""
        return int(self.lib.mgn_halo_bytes_per_row(self.h)) // 4

    def halo_pack(self, send_ptr):
        self._chk(self.lib.mgn_halo_pack(self.h, C.c_void_p(send_ptr)))

    def halo_unpack(self, recv_ptr):
        self._chk(self.lib.mgn_halo_unpack(self.h, C.c_void_p(recv_ptr)))

    # -- communicator: the halo exchange inside the library ---------------------
    @staticmethod
    def comm_unique_id(transport="rccl"):
        """MGN_COMM_ID_BYTES bytes made on ONE rank; the host distributes them (a torch.distributed store, MPI, a file)."""
        lib = _capi.load()
        buf = C.create_string_buffer(_capi.MGN_COMM_ID_BYTES)
        rc = lib.mgn_comm_unique_id(buf, {"rccl": 0, "host": 1}[transport])
        if rc != 0:
            raise MgnError(rc, lib.mgn_last_error(None).decode())
        return buf.raw

    def comm_init(self, comm_id, transport="rccl"):
        """Collective over the nranks handles of the partitioned mesh; afterwards processor_steps_dev / forward run at
        nranks > 1 with the halo exchange inside the library.  transport "rccl" (one GPU per rank) or "host" (shared memory)."""
        if len(comm_id) != _capi.MGN_COMM_ID_BYTES:
            raise ValueError("comm_id must be MGN_COMM_ID_BYTES bytes")
        buf = C.create_string_buffer(bytes(comm_id), _capi.MGN_COMM_ID_BYTES)
        self._chk(self.lib.mgn_comm_init(self.h, buf, _capi.MGN_COMM_ID_BYTES, {"rccl": 0, "host": 1}[transport]))

    def comm_init_file(self, path, transport="rccl"):
        self._chk(self.lib.mgn_comm_init_file(self.h, str(path).encode(), {"rccl": 0, "host": 1}[transport]))

    def comm_destroy(self):
        self._chk(self.lib.mgn_comm_destroy(self.h))

    def comm_barrier(self):
        self._chk(self.lib.mgn_comm_barrier(self.h))

    def comm_allreduce(self, x, op="sum"):
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        self._chk(self.lib.mgn_comm_allreduce(self.h, x.ctypes.data_as(C.POINTER(C.c_double)), x.size, {"sum": 0, "max": 1}[op]))
        return x

    def halo_exchange(self):
        self._chk(self.lib.mgn_halo_exchange(self.h))

    def halo_exchange_host(self, own_rows):
        """per-node host rows [n_own][W] -> rows of this partition's halo nodes [n_halo][W] (order of halo_nodes())"""
        own_rows = _c32(own_rows)
        if own_rows.ndim != 2 or own_rows.shape[0] != self.n_own:
            raise ValueError("DimensionMismatch: own_rows must be [n_own][W]")
        out = np.zeros((self.n_halo, own_rows.shape[1]), np.float32)
        self._chk(self.lib.mgn_halo_exchange_host(self.h, f32(own_rows), f32(out), own_rows.shape[1]))
        return out

    # -- measurement ---------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self.lib.mgn_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        ms = (C.c_double * 8)()
        cnt = (C.c_int64 * 8)()
        self._chk(self.lib.mgn_profile_read(self.h, ms, cnt))
        names = ["edge_step", "node_step", "encode", "decode", "halo", "edge_boundary"]
        return {n: dict(avg_ms=ms[i], count=cnt[i]) for i, n in enumerate(names)}


def triangles_to_edges_native(cells):
    """GraphNetCore.triangles_to_edges at scale (C++ sort/unique on packed keys): returns (senders, receivers)."""
    lib = _capi.load()
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    if cells.ndim != 2 or cells.shape[1] != 3:
        raise ValueError("DimensionMismatch: cells must be [C][3]")
    n = C.c_int64()
    rc = lib.mgn_triangles_to_edges(i32(cells), cells.shape[0], None, None, C.byref(n))
    if rc != 0:
        raise MgnError(rc, "mgn_triangles_to_edges")
    s, r = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
    rc = lib.mgn_triangles_to_edges(i32(cells), cells.shape[0], i32(s), i32(r), C.byref(n))
    if rc != 0:
        raise MgnError(rc, "mgn_triangles_to_edges")
    return s, r


def world_edges_native(world_pos, radius, mesh_senders, mesh_receivers, index_base=0):
    """Radius graph in world space minus self loops and mesh-edge pairs (world edges of cloth models): (senders, receivers)."""
    lib = _capi.load()
    wp = _c32(world_pos)
    ms = np.ascontiguousarray(mesh_senders, dtype=np.int32)
    mr = np.ascontiguousarray(mesh_receivers, dtype=np.int32)
    n = C.c_int64()
    args = (f32(wp), wp.shape[1], wp.shape[0], C.c_float(radius), i32(ms), i32(mr), ms.size, index_base)
    rc = lib.mgn_world_edges(*args, None, None, C.byref(n))
    if rc != 0:
        raise MgnError(rc, "mgn_world_edges")
    s, r = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
    rc = lib.mgn_world_edges(*args, i32(s), i32(r), C.byref(n))
    if rc != 0:
        raise MgnError(rc, "mgn_world_edges")
    return s, r


def edge_features_native(mesh_pos, senders, receivers, index_base=0):
    lib = _capi.load()
    pos = _c32(mesh_pos)
    s = np.ascontiguousarray(senders, dtype=np.int32)
    r = np.ascontiguousarray(receivers, dtype=np.int32)
    ef = np.empty((s.size, pos.shape[1] + 1), np.float32)
    rc = lib.mgn_edge_features(f32(pos), pos.shape[1], i32(s), i32(r), s.size, index_base, f32(ef))
    if rc != 0:
        raise MgnError(rc, "mgn_edge_features")
    return ef


# ==================================================================================================
# GraphNetCore-shaped surface
# ==================================================================================================
class FeatureGraph:
    """FeatureGraph(nf, ef, senders, receivers) -- reference src/graph.jl:87-96.  Single edge set."""

    def __init__(self, nf, ef, senders, receivers):
        self.nf, self.ef, self.senders, self.receivers = nf, ef, senders, receivers


class _Model:
    """Callable `mgn.model(graph, ps, st) -> (output, st)` (reference src/solve.jl:200)."""

    def __init__(self, net):
        self.net = net

    def __call__(self, graph, ps, st):
        net = self.net
        net._sync_params(ps)
        net._sync_graph(graph.senders, graph.receivers, np.asarray(graph.nf).shape[0])
        return net.engine.forward(graph.nf, graph.ef), st


class GraphNetwork:
    """Mutable holder mirroring GraphNetCore.GraphNetwork: fields model, ps, st, e_norm, n_norm, o_norm
    (reference reads/writes them at src/solve.jl:54,200-208, src/graph.jl:80-93,
    src/MeshGraphNets.jl:288,376-377).  `ps` is the packed float32 parameter vector (MGN-spec order),
    owned by the caller so an optimiser can update it in place; it is re-uploaded when it changes."""

    def __init__(self, quantities, dims, e_norm, n_norm, o_norm, outputs, mps=15, layer_size=128,
                 hidden_layers=2, ps=None, index_base=0, device=-1):
        self.engine = Engine(quantities, dims + 1, outputs, layer_size, hidden_layers, mps, device=device)
        self.e_norm, self.n_norm, self.o_norm = e_norm, n_norm, o_norm
        self.st = None
        self.index_base = index_base
        if ps is None:
            raise ValueError("ps (packed parameters) is required: use load_params or init_params")
        self.ps = np.ascontiguousarray(ps, np.float32)
        self._ps_token = None
        self._graph_token = None
        self.model = _Model(self)

    def _sync_params(self, ps):
        ps = np.ascontiguousarray(ps, np.float32)
        tok = (ps.ctypes.data, ps.size, float(ps[:64].sum()), float(ps[-64:].sum()), float(ps[::4099].sum()))
        if tok != self._ps_token:
            self.engine.set_params(ps)
            self._ps_token = tok

    def _sync_graph(self, senders, receivers, N):
        s = np.asarray(senders)
        tok = (s.ctypes.data if s.flags.c_contiguous else id(senders), s.size, N, int(s[:16].sum()) if s.size else 0)
        if tok != self._graph_token:
            self.engine.set_graph(senders, receivers, N, index_base=self.index_base)
            self._graph_token = tok

    def set_graph(self, senders, receivers, N, mesh_pos=None):
        """Explicit once-per-trajectory call (what create_base_graph amortises, src/MeshGraphNets.jl:360)."""
        self.engine.set_graph(senders, receivers, N, index_base=self.index_base, mesh_pos=mesh_pos)
        s = np.asarray(senders)
        self._graph_token = (s.ctypes.data if s.flags.c_contiguous else id(senders), s.size, N,
                             int(s[:16].sum()) if s.size else 0)


def init_params(cfg, seed=None):
    """Fresh parameters in packed order: Glorot-uniform weights, zero biases, LayerNorm scale 1 / bias 0 (the twin of
    MGNHip.init_params; NumPy's draws).  cfg: (Fn, Fe, O, L, hidden_layers, mps)."""
    Fn, Fe, O, L, h, mps = (int(x) for x in cfg)
    rng = np.random.default_rng(seed)
    parts = []

    def mlp(fin, fout, ln):
        dims = [fin] + [L] * h + [fout]
        for a, b in zip(dims[:-1], dims[1:]):
            lim = np.sqrt(6.0 / (a + b))
            parts.append(((rng.random(a * b, dtype=np.float32) * 2 - 1) * lim).astype(np.float32))
            parts.append(np.zeros(b, np.float32))
        if ln:
            parts.extend((np.ones(fout, np.float32), np.zeros(fout, np.float32)))

    mlp(Fn, L, True)
    mlp(Fe, L, True)
    for _ in range(mps):
        mlp(3 * L, L, True)
        mlp(2 * L, L, True)
    mlp(L, O, False)
    return np.concatenate(parts)


def load(quantities, dims, e_norms, n_norms, o_norms, outputs, mps, layer_size, hidden_layers, opt, device, path, index_base=0,
         hip_device=-1, seed=None):
    """`load(...) -> (mgn, opt_state, df_train, df_valid)` with the twelve positional arguments of the reference's call sites
    (src/MeshGraphNets.jl:282-285, 537-540).  With a checkpoint written by `save` in `path`, ALL of its state comes back: parameters,
    loss log, the normalisers' statistics restored over the freshly built `e_norms` / `n_norms` / `o_norms` (what `eval_network`
    relies on, :529-540) and `opt_state` (kept on resume, :287-289; None without a checkpoint or when `opt` is None).  `device` (the
    reference's Lux device function) is accepted and ignored."""
    from . import checkpoint as ck
    probe = _capi.load().mgn_param_count
    cfg = MgnConfig(Fn=quantities, Fe=dims + 1, O=outputs, L=layer_size, hidden_layers=hidden_layers, mps=mps, n_edge_sets=1)
    nparams = int(probe(C.byref(cfg)))
    got = ck.read_checkpoint(path, nparams, e_norms, n_norms, o_norms, want_opt_state=opt is not None)
    if got is None:
        ps, opt_state, df_train, df_valid = init_params((quantities, dims + 1, outputs, layer_size, hidden_layers, mps), seed), None, ck.LossLog(), ck.LossLog()
    else:
        ps, e_norms, n_norms, o_norms, opt_state, df_train, df_valid = got
    mgn = GraphNetwork(quantities, dims, e_norms, n_norms, o_norms, outputs, mps, layer_size, hidden_layers, ps=ps,
                       index_base=index_base, device=hip_device)
    return mgn, opt_state, df_train, df_valid


def save(mgn, opt_state, df_train, df_valid, step, loss, path, is_training=True):
    """`save!(mgn, opt_state, df_train, df_valid, step, loss, path; is_training)` (src/MeshGraphNets.jl:460-471): appends (step, loss)
    to the training or validation log and writes parameters, normalisers (with whatever an online one has accumulated), optimiser
    state and the log."""
    from . import checkpoint as ck
    log = df_train if is_training else df_valid
    log.step.append(int(step))
    log.loss.append(np.float32(loss))
    ck.write_checkpoint(path, mgn.ps, mgn.e_norm, mgn.n_norm, mgn.o_norm, opt_state, df_train, df_valid)


def step(mgn, graph, target, mask, loss_function=None):
    """GraphNetCore.step!(mgn, graph, target, mask, loss_function) as the reference calls it (src/strategies.jl:418-422):
    returns (gs, loss) with loss = mean(mse_reduce(target, mgn.model(graph))[mask]) and gs = d loss / d mgn.ps in packed
    order, ready for the optimiser update at src/MeshGraphNets.jl:375-377.  `mask` follows mgn.index_base (1-based Int32 node
    indices at the Julia boundary).  Only the reference's loss (mse_reduce) exists on the device: any other callable
    is refused rather than silently replaced."""
    if loss_function is not None and getattr(loss_function, "__name__", "") != "mse_reduce":
        raise ValueError("step: only mse_reduce is implemented on the device (reference src/strategies.jl:421)")
    mgn._sync_params(mgn.ps)
    mgn._sync_graph(graph.senders, graph.receivers, graph.nf.shape[0])
    return mgn.engine.step(graph.nf, graph.ef, target, mask, mask_index_base=mgn.index_base)


# ==================================================================================================
# staged driver shared by the single-GPU loopback test, the RCCL path and the gloo CPU test
# ==================================================================================================
def run_processor_staged(engines, exchange, nsteps, begin=True, overlap=None):
    """engines: objects with proc_begin/proc_edge/proc_node (one per local partition);
    exchange(): performs pack -> all-to-all-v -> unpack for all of them.
    Mirrors mgn_processor_steps_dev for nranks > 1.

    overlap (default: on when the engines and the exchange support it): owned nodes are numbered boundary-first,
    so the projection of the boundary tiles runs first, the exchange is started (exchange.start(): pack + async
    all-to-all-v) and the interior tiles are projected while the rows are on the wire; the next edge step then runs
    its interior tiles (no halo sender) first and only its few boundary tiles after exchange.finish()."""
    if nsteps <= 0:
        return
    if overlap is None:
        overlap = all(hasattr(e, "proc_node_phase") for e in engines) and hasattr(exchange, "start")

    edge_split = overlap and all(hasattr(e, "proc_edge_phase") for e in engines)

    def project_and_start(k):           # k = -1: projection for step 0 (proc_begin); leaves the exchange in flight
        if overlap:
            for e in engines:
                e.proc_node_phase(k, 1)
            exchange.start()
            for e in engines:
                e.proc_node_phase(k, 2)
        else:
            for e in engines:
                e.proc_begin() if k < 0 else e.proc_node(k, True)
            exchange.start() if hasattr(exchange, "start") else exchange()

    def finish():
        if hasattr(exchange, "finish"):
            exchange.finish()

    if begin:
        project_and_start(-1)
    else:
        exchange.start() if hasattr(exchange, "start") else exchange()
    for k in range(nsteps):
        if edge_split:
            # edges whose sender is owned run while the halo rows are on the wire; the few boundary tiles after them
            for e in engines:
                e.proc_edge_phase(k, 1)
            finish()
            for e in engines:
                e.proc_edge_phase(k, 2)
        else:
            finish()
            for e in engines:
                e.proc_edge(k)
        if k + 1 < nsteps:
            project_and_start(k)
        else:
            for e in engines:
                e.proc_node(k, False)


def run_forward_staged(engines, exchange, mps):
    for e in engines:
        e.fwd_encode()
    run_processor_staged(engines, exchange, mps, begin=False)
    for e in engines:
        e.fwd_decode()
