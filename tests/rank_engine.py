"""NumPy stand-in for one partition's engine (TEST INFRASTRUCTURE): same staged interface as mgn_amd.Engine
(proc_begin / proc_edge / proc_node / halo_*), arithmetic from the float64 oracle's parameters, written in the
FACTORED form the HIP kernels use (P = v W1s, Q = v W1r + b1, edge layer 1 = P[s] + Q[r] + e W1e), partition
data from the C library's host-only handle.  Used by the gloo world_size-2 test and the loopback test."""
import numpy as np

import mgn_amd
import mgn_oracle as orc


class OracleRankEngine:
    def __init__(self, cfg, ps, senders, receivers, N, mesh_pos, rank, nranks):
        self.cfg = cfg
        self.h = cfg["hidden_layers"]
        self.P_ = orc.unpack_params(np.asarray(ps, np.float64), cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], self.h, cfg["mps"])
        self.part = mgn_amd.Engine(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], self.h, cfg["mps"], rank=rank, nranks=nranks,
                                   device=mgn_amd.MGN_DEVICE_NONE)
        self.part.set_graph(senders, receivers, N, mesh_pos=mesh_pos)
        self.own = self.part.owned_nodes()
        self.eid = self.part.local_edges()
        self.snd, self.rcv, self.rowptr = self.part.local_graph()
        self.send_idx = self.part.halo_send_index()
        self.n_own, self.n_halo = self.part.n_own, self.part.n_halo
        self.n_bnd_rows = min(self.n_own, -(-self.part.boundary_count() // 32) * 32)   # whole boundary tiles
        L = cfg["L"]
        self.V = np.zeros((self.n_own, L))
        self.E = np.zeros((self.eid.size, L))
        self.P = np.zeros((self.n_own + self.n_halo, L))
        self.Q = np.zeros((self.n_own, L))
        self.halo_row_floats = L

    def halo_counts(self):
        return self.part.halo_counts()

    def latents_import(self, v, e):
        self.V = np.asarray(v, np.float64)[self.own].copy()
        self.E = np.asarray(e, np.float64)[self.eid].copy()

    def latents_export(self, v, e):
        v[self.own] = self.V
        e[self.eid] = self.E

    def _project(self, k, lo=0, hi=None):
        W1, b1 = self.P_["proc%d_edge" % k]["W1"], self.P_["proc%d_edge" % k]["b1"]
        L = self.cfg["L"]
        hi = self.n_own if hi is None else hi
        self.P[lo:hi] = self.V[lo:hi] @ W1[0:L]
        self.Q[lo:hi] = self.V[lo:hi] @ W1[L:2 * L] + b1

    def proc_node_phase(self, k, phase):
        """phase 1: node MLP of step k (k >= 0) + projection of the boundary tiles; phase 2: interior tiles."""
        if phase == 1:
            if k >= 0:
                self.proc_node(k, False)
            self._project(k + 1, 0, self.n_bnd_rows)
        else:
            self._project(k + 1, self.n_bnd_rows, self.n_own)

    def proc_begin(self):
        self._project(0)

    def _edge_range(self, k, lo, hi):
        p = self.P_["proc%d_edge" % k]
        L = self.cfg["L"]
        sl = slice(lo, hi)
        h1 = np.maximum(self.P[self.snd[sl]] + self.Q[self.rcv[sl]] + self.E[sl] @ p["W1"][2 * L:], 0.0)
        h2 = np.maximum(h1 @ p["W2"] + p["b2"], 0.0)
        en = orc.layer_norm(h2 @ p["W3"] + p["b3"], p["ln_scale"], p["ln_bias"])
        self.agg += orc.scatter_add(en, self.rcv[sl], self.n_own)
        self.E[sl] = self.E[sl] + en

    def proc_edge(self, k):
        self.agg = np.zeros((self.n_own, self.cfg["L"]))
        self._edge_range(k, 0, self.eid.size)

    def proc_edge_phase(self, k, phase):
        """phase 1: the edge tiles without halo senders (runs BEFORE the halo rows have arrived: they are poisoned with
        NaN until halo_unpack_tensor, so a misplaced edge would show); phase 2: the boundary tiles."""
        tb, _ = self.part.edge_boundary_tiles()
        cut = min(tb * 32, self.eid.size)
        if phase == 1:
            self.agg = np.zeros((self.n_own, self.cfg["L"]))
            self._edge_range(k, cut, self.eid.size)
        else:
            self._edge_range(k, 0, cut)

    def proc_node(self, k, project_next):
        vn = orc.mlp(np.concatenate([self.V, self.agg], 1), self.P_["proc%d_node" % k], self.h)
        self.V = self.V + vn
        if project_next:
            self._project(k + 1)

    def halo_pack_tensor(self, t):
        import torch
        if self.send_idx.size:
            t[: self.send_idx.size] = torch.from_numpy(self.P[self.send_idx].astype(np.float32))
        self.P[self.n_own:] = np.nan          # stale halo rows must not be read before the exchange has finished

    def halo_unpack_tensor(self, t):
        if self.n_halo:
            self.P[self.n_own:] = t[: self.n_halo].cpu().numpy().astype(np.float64)
