"""Host-side mirror of the reference functions on either side of the hot path, with the reference's names.

In the real drop-in these stay Julia and unchanged (reference src/graph.jl, src/solve.jl call into
GraphNetCore; only GraphNetCore is replaced).  Julia is not available in this environment, so the same
functions are written here in NumPy float32 to drive the engine exactly the way the reference does:

    create_base_graph      reference src/graph.jl:25-55
    build_graph            reference src/graph.jl:75-97
    ode_func_eval          reference src/solve.jl:147-158
    ode_step               reference src/solve.jl:188-219
    rollout (Euler branch) reference src/solve.jl:42-68  (`solve(prob, solver; adaptive=false, dt, saveat)`)
    GraphNetCore surface   one_hot, triangles_to_edges, parse_edges, mse_reduce, NormaliserOfflineMinMax,
                           NormaliserOfflineMeanStd, NormaliserOnline, inverse_data (docs/src/graph_net_core.md)

Arrays are [count][feat] (the bytes of Julia's feat x count).  Indices returned by triangles_to_edges /
parse_edges are 0-based here; GraphNetwork(index_base=...) tells the engine which base the caller uses.
"""
from __future__ import annotations

import os

import numpy as np

from .engine import FeatureGraph

F32 = np.float32


# ---- GraphNetCore utilities ---------------------------------------------------------------------
def one_hot(v, depth, offset=0):
    """one_hot(vec(node_type), type_max - type_min + 1, 1 - type_min) at src/graph.jl:26-27 (the Julia offset
    is 1-based; here offset = -type_min)."""
    v = np.asarray(v).reshape(-1).astype(np.int64) + offset
    if v.size and (v.min() < 0 or v.max() >= depth):
        raise ValueError("ArgumentError: one_hot index outside [0, depth)")
    out = np.zeros((v.size, depth), F32)
    out[np.arange(v.size), v] = 1.0
    return out


def triangles_to_edges(cells):
    """Two-way unique edges of a triangulation in first-occurrence order (src/graph.jl:30)."""
    cells = np.asarray(cells, dtype=np.int64)
    if cells.ndim != 2 or cells.shape[1] != 3:
        raise ValueError("DimensionMismatch: cells must be [C][3]")
    e = np.concatenate([cells[:, 0:2], cells[:, 1:3], cells[:, [2, 0]]], 0)
    hi, lo = e.max(1), e.min(1)
    key = hi * (int(e.max()) + 1) + lo
    _, first = np.unique(key, return_index=True)
    first.sort()
    a, b = hi[first].astype(np.int32), lo[first].astype(np.int32)
    return np.concatenate([a, b]), np.concatenate([b, a])


def parse_edges(edges):
    """parse_edges(data["edges"]) at src/graph.jl:38: explicit [n][2] edge list -> two-way senders/receivers."""
    e = np.asarray(edges, dtype=np.int32)
    if e.ndim != 2 or e.shape[1] != 2:
        raise ValueError("DimensionMismatch: edges must be [n][2]")
    return np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]])


def mse_reduce(target, output):
    """Sum of squared errors over the feature rows, per node (loss_fn of step!, src/strategies.jl:421)."""
    return ((np.asarray(target, F32) - np.asarray(output, F32)) ** 2).sum(-1)


# ---- normalisers (constructed at src/MeshGraphNets.jl:79-203) --------------------------------------
class NormaliserOfflineMinMax:
    def __init__(self, data_min, data_max, target_min=0.0, target_max=1.0):
        self.data_min, self.data_max = F32(data_min), F32(data_max)
        self.target_min, self.target_max = F32(target_min), F32(target_max)

    def __call__(self, x):
        x = np.asarray(x, F32)
        return (x - self.data_min) / (self.data_max - self.data_min) * (self.target_max - self.target_min) + self.target_min

    def inverse(self, y):
        y = np.asarray(y, F32)
        return (y - self.target_min) / (self.target_max - self.target_min) * (self.data_max - self.data_min) + self.data_min

    def affine(self, dim):
        s = (self.target_max - self.target_min) / (self.data_max - self.data_min)
        return np.full(dim, s, F32), np.full(dim, self.target_min - self.data_min * s, F32)

    def inverse_affine(self, dim):
        s = (self.data_max - self.data_min) / (self.target_max - self.target_min)
        return np.full(dim, s, F32), np.full(dim, self.data_min - self.target_min * s, F32)


class NormaliserOfflineMeanStd:
    def __init__(self, mean, std):
        self.mean = np.asarray(mean, F32)
        self.std = np.maximum(np.asarray(std, F32), F32(1e-8))

    def __call__(self, x):
        return (np.asarray(x, F32) - self.mean) / self.std

    def inverse(self, y):
        return np.asarray(y, F32) * self.std + self.mean

    def affine(self, dim):
        s = np.broadcast_to(F32(1.0) / self.std, (dim,)).astype(F32)
        return s, (-np.broadcast_to(self.mean, (dim,)) * s).astype(F32)

    def inverse_affine(self, dim):
        return np.broadcast_to(self.std, (dim,)).astype(F32), np.broadcast_to(self.mean, (dim,)).astype(F32)


class NormaliserOnline:
    """NormaliserOnline(dims, device; max_acc): accumulates sum / sum of squares / count over the rows it is
    called with until max_acc calls, then normalises with the running mean / max(std, 1e-8)."""

    def __init__(self, dims, device=None, max_acc=1e6, std_epsilon=1e-8):
        self.dims, self.max_acc, self.std_epsilon = int(dims), float(max_acc), F32(std_epsilon)
        self.acc_sum = np.zeros(self.dims, np.float64)
        self.acc_sum_squared = np.zeros(self.dims, np.float64)
        self.acc_count = 0.0
        self.num_accumulations = 0.0
        self.engine = None               # set to an Engine to accumulate on the device

    def _accumulate(self, x):
        if self.num_accumulations < self.max_acc:
            if self.engine is not None:          # device reduction (mgn_feature_stats), same float64 totals
                s, q = self.engine.feature_stats(np.ascontiguousarray(x, dtype=np.float32))
                self.acc_sum += s
                self.acc_sum_squared += q
            else:
                self.acc_sum += x.sum(0, dtype=np.float64)
                self.acc_sum_squared += (x.astype(np.float64) ** 2).sum(0)
            self.acc_count += x.shape[0]
            self.num_accumulations += 1.0

    def frozen(self):
        c = max(self.acc_count, 1.0)
        mean = self.acc_sum / c
        std = np.sqrt(np.maximum(self.acc_sum_squared / c - mean * mean, 0.0))
        return NormaliserOfflineMeanStd(mean.astype(F32), np.maximum(std, self.std_epsilon).astype(F32))

    def __call__(self, x, accumulate=True):
        x = np.asarray(x, F32)
        if x.shape[-1] != self.dims:
            raise ValueError("DimensionMismatch: NormaliserOnline built for %d features, got %d" % (self.dims, x.shape[-1]))
        if accumulate:
            self._accumulate(x)
        return self.frozen()(x)

    def inverse(self, y):
        return self.frozen().inverse(y)

    def affine(self, dim):
        return self.frozen().affine(dim)

    def inverse_affine(self, dim):
        return self.frozen().inverse_affine(dim)


def inverse_data(norm, y):
    """inverse_data(mgn.o_norm[field], output_rows) at src/solve.jl:207."""
    return norm.inverse(y)


# ---- src/graph.jl ------------------------------------------------------------------------------------
def create_base_graph(data, type_size, type_min, device=None):
    """Static part of the graph, once per trajectory.  data: dict with 'node_type' [N] (or [N][1]), 'mesh_pos'
    [N][dims] and 'cells' [C][3] or 'edges' [n][2] (either index base, told apart as the reference does: by the presence of a 0).
    Per-node arrays may carry a leading time axis ([T][N][..], what the dataset readers return): the first frame is taken, like
    `data[...][:, :, 1]`.  Returns (node_type_onehot, senders, receivers, edge_features) like src/graph.jl:54 (0-based indices)."""
    first = lambda a, nd: np.asarray(a)[0] if np.asarray(a).ndim > nd else np.asarray(a)     # noqa: E731  ([T][N][..] -> frame 1)
    nt = np.asarray(data["node_type"])
    nt = nt[0] if nt.ndim == 3 else nt
    node_type = one_hot(nt.reshape(-1), type_size - type_min + 1, -type_min)
    if "cells" in data:
        senders, receivers = triangles_to_edges(first(data["cells"], 2))
    elif "edges" in data:
        senders, receivers = parse_edges(data["edges"])
    else:
        raise KeyError("Data does not contain cell or edge information!")
    # src/graph.jl:31-34, 39-42: lists that name a node 0 are 0-based and shifted to Julia's 1-based indices, all others are taken as 1-based
    # already (the HDF5 arm's create_edges / read_edges produce those).  This module indexes from 0: the mirror image of that rule.
    if senders.size and not ((senders == 0).any() or (receivers == 0).any()):
        senders, receivers = senders - 1, receivers - 1
    pos = np.asarray(first(data["mesh_pos"], 2), F32)
    rel = pos[senders] - pos[receivers]
    edge_features = np.concatenate([rel, np.linalg.norm(rel, axis=1, keepdims=True).astype(F32)], 1).astype(F32)
    return node_type, senders, receivers, edge_features


def build_graph(mgn, data, fields, datapoint, node_type, edge_features, senders, receivers):
    """nf = [n_norm[field](data[field]) for field in fields ..., n_norm['node_type'](node_type)]; ef = e_norm(...)
    (src/graph.jl:80-96).  data[field]: [N][dim] or [T][N][dim] (then row min(T, datapoint) is used)."""
    if np.asarray(edge_features).dtype != np.float32:
        raise TypeError("MethodError: edge_features must be Float32 (src/graph.jl:76)")
    cols = []
    for field in fields:
        d = np.asarray(data[field], F32)
        if d.ndim == 3:
            d = d[min(d.shape[0] - 1, datapoint)]
        cols.append(mgn.n_norm[field](d))
    cols.append(mgn.n_norm["node_type"](node_type))
    return FeatureGraph(np.concatenate(cols, 1).astype(F32), mgn.e_norm(edge_features).astype(F32), senders, receivers)


# ---- src/solve.jl ------------------------------------------------------------------------------------
def ode_step(x, p, t):
    """x: [N][sum(dims)] state; p = (mgn, ps, inputs, fields, meta, target_fields, target_dict, node_type,
    edge_features, senders, receivers, val_mask, pr) exactly as src/solve.jl:188-191."""
    (mgn, ps, inputs, fields, meta, target_fields, target_dict, node_type, edge_features, senders, receivers,
     val_mask, pr) = p
    offset = 0
    for k in target_fields:
        inputs[k] = x[:, offset:offset + target_dict[k]]
        offset += target_dict[k]
    graph = build_graph(mgn, inputs, fields, 0, node_type, edge_features, senders, receivers)
    output, st = mgn.model(graph, ps, mgn.st)
    mgn.st = st
    indices = [meta["features"][tf]["dim"] for tf in target_fields]
    buf = np.empty_like(output)
    o = 0
    for i, tf in enumerate(target_fields):
        buf[:, o:o + indices[i]] = inverse_data(mgn.o_norm[tf], output[:, o:o + indices[i]])
        o += indices[i]
    return buf * val_mask


def ode_func_eval(x, p, t):
    """Inflow overwrite then ode_step (src/solve.jl:147-158).  p as in the reference plus (data, inflow_mask,
    saves_dt): x[inflow_mask] = vcat(data[field][floor(t / saves_dt)] ...)[inflow_mask]."""
    (mgn, ps, data, inputs, fields, meta, target_fields, target_dict, node_type, edge_features, senders, receivers,
     val_mask, inflow_mask, saves_dt, pr) = p
    # floor(Int, t / saves_dt) + 1 (1-based) in the type t and saves_dt arrive in, no tolerance (src/solve.jl:151): an IndexError
    # here is the reference's BoundsError
    k = int(np.floor(t / saves_dt))
    gt = np.concatenate([np.asarray(data[f], F32)[k] for f in target_fields], 1)
    x[inflow_mask] = gt[inflow_mask]   # IN PLACE, like the reference: the caller's (solver's) state is modified
    return ode_step(x, (mgn, ps, inputs, fields, meta, target_fields, target_dict, node_type, edge_features,
                        senders, receivers, val_mask, pr), t)


def rollout(solver, mgn, initial_state, fields, meta, target_fields, target_dict, node_type, edge_features, senders,
            receivers, val_mask, inflow_mask, data, start, stop, dt, saves, show_progress=False):
    """Fixed-step branch of rollout (src/solve.jl:57-61, `adaptive = false, dt = dt`); solver must be "Euler".
    Returns (sol_u [len(saves)][N][O], sol_t)."""
    if solver != "Euler" or dt is None:
        raise NotImplementedError("only the fixed-step Euler branch is mirrored; adaptive Tsit5 stays in DifferentialEquations.jl")
    x = np.concatenate([np.asarray(initial_state[f], F32) for f in target_fields], 1)
    inputs = {k: v for k, v in initial_state.items() if k not in target_dict}
    p = (mgn, mgn.ps, data, inputs, fields, meta, target_fields, target_dict, node_type, edge_features, senders,
         receivers, val_mask, inflow_mask, saves[1] - saves[0], None)
    sol_u, sol_t = [], []
    t = start
    nsteps = int(round((stop - start) / dt))
    save_set = {int(round((s - start) / dt)) for s in saves}
    for i in range(nsteps + 1):
        if i in save_set:
            sol_u.append(x.copy())
            sol_t.append(t)
        if i == nsteps:
            break
        dx = ode_func_eval(x, p, t)   # mutates the inflow rows of x (reference quirk, src/solve.jl:151-152)
        x = x + F32(dt) * dx
        t = t + dt                    # the integrator's own time: t <- t + dt in the type start / dt arrive in (np.float32 or float)
    return np.stack(sol_u), np.array(sol_t)


# ---- dataset formats -------------------------------------------------------------------------------
class TFRecordReader:
    """Iterator over the tf.train.Example records of a .tfrecord file (native reader: mgn_tfrecord_*), the stand-in for
    TFRecord.jl's `read(path; channel_size)` (reference src/dataset.jl:107-112).  Yields {feature name: (kind, bytes)}
    with kind 1 = bytes_list (first value), 2 = float_list, 3 = int64_list."""

    def __init__(self, path, verify_crc=True):
        import ctypes as C
        from . import _capi
        self._C, self.lib = C, _capi.load()
        self.h = C.c_void_p()
        if self.lib.mgn_tfrecord_open(str(path).encode(), 1 if verify_crc else 0, C.byref(self.h)) != 0:
            raise FileNotFoundError(path)

    def __iter__(self):
        return self

    def __next__(self):
        C = self._C
        rc = self.lib.mgn_tfrecord_next(self.h)
        if rc == 0:
            raise StopIteration
        if rc < 0:
            raise ValueError("ArgumentError: " + self.lib.mgn_tfrecord_error(self.h).decode())
        out = {}
        for i in range(self.lib.mgn_tfrecord_feature_count(self.h)):
            name = self.lib.mgn_tfrecord_feature_name(self.h, i)
            kind, ptr, nb = C.c_int32(), C.c_void_p(), C.c_int64()
            self.lib.mgn_tfrecord_feature(self.h, name, C.byref(kind), C.byref(ptr), C.byref(nb))
            out[name.decode()] = (kind.value, C.string_at(ptr, nb.value) if nb.value else b"")
        return out

    def close(self):
        if self.h:
            self.lib.mgn_tfrecord_close(self.h)
            self.h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_data(example, meta):
    """parse_data(data::Example, meta) (reference src/dataset.jl:61-75): every feature listed in meta["features"] is
    reinterpreted by its dtype and reshaped to reverse(shape) (a -1 entry resolved from the payload length); static
    features are repeated over the trajectory.  Arrays come back as [T][count][dim] (the bytes of Julia's dim x count x T).
    A feature missing from the record raises KeyError, like `data.features.feature[key]`."""
    out = {}
    for key, value in meta["features"].items():
        kind, payload = example[key]
        d = np.frombuffer(payload, dtype=np.dtype(value["dtype"]))
        shape = list(value["shape"])
        if -1 in shape:
            q = d.size
            for sdim in shape:                      # abs(reduce(div, shape; init = length(d)))
                q = int(q / sdim) if sdim else q    # Julia `div` truncates toward zero
            shape[shape.index(-1)] = abs(q)
        d = d.reshape(shape)
        if value["type"] == "static":
            d = np.repeat(d, meta["trajectory_length"], axis=0)
        out[key] = d
    return out


class Dataset:
    """Dataset (reference src/dataset.jl:36-47): file, file_valid, meta, ch, ch_valid.  `ch` / `ch_valid` are iterators standing in for
    the Channels; unpacking gives (meta, ch), the pair the TFRecord-only version of load_dataset returned."""

    def __init__(self, file, file_valid, meta, ch, ch_valid):
        self.file, self.file_valid, self.meta, self.ch, self.ch_valid = file, file_valid, meta, ch, ch_valid

    def __iter__(self):
        return iter((self.meta, self.ch))


def load_dataset(path, is_training):
    """load_dataset(path, is_training) (reference src/dataset.jl:89-172): `train` or `test` + `.tfrecord`, else `.jld2`, else `.h5`,
    in that order of preference; training also opens the `valid` file of the same format.  The TFRecord arm yields parsed trajectories
    (parse_data), the HDF5 / JLD2 arm the dictionaries of read_h5! (`dataset_h5.read_trajectory`; needs libhdf5, `hdf5_lite`)."""
    import json
    import os
    filename = "train" if is_training else "test"
    if os.path.isfile(os.path.join(path, filename + ".tfrecord")):
        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        file, valid = os.path.join(path, filename + ".tfrecord"), os.path.join(path, "valid.tfrecord")

        def records(f):                       # opened when first taken from, like the task behind TFRecord.jl's Channel
            for ex in TFRecordReader(f):
                yield parse_data(ex, meta)
        return Dataset(file, valid, meta, records(file), records(valid) if is_training else None)
    file = filename + (".jld2" if os.path.isfile(os.path.join(path, filename + ".jld2")) else ".h5")
    if not os.path.isfile(os.path.join(path, file)):
        raise FileNotFoundError(f"ArgumentError: {path} does not contain a {filename}.tfrecord or a {filename}.h5 file")
    from . import dataset_h5
    meta, ch, ch_valid = dataset_h5.load_dataset_h5(path, is_training, file)
    return Dataset(os.path.join(path, file), os.path.join(path, "valid" + os.path.splitext(file)[1]), meta, ch, ch_valid)


# ---- trajectory preparation (host side of SURVEY.md N3 / N4; arrays are [T][count][dim] like parse_data's) -------------
def dump_rollout(dump_dir, ti, mesh_pos, gt, prediction, error, timesteps, cells=None):
    """The evaluation output of one trajectory (reference src/MeshGraphNets.jl:630-637: traj_ops / errors / timesteps / cells of
    eval_network!) as a raw dump that julia/write_trajectories.jl turns into the reference's `trajectories.h5`
    (`/<ti>/<name>/{data,size}`, src/MeshGraphNets.jl:638-669) -- for hosts without libhdf5; with it, `dataset_h5.write_trajectories_h5`
    writes the file directly.
    Arrays arrive in this module's [time][count][feat] / [count][feat] order and are written as the bytes of the Julia arrays
    (feat x count x time: the same bytes); `size` is the Julia size.  ti is 1-based like the reference's trajectory counter."""
    import json
    d = os.path.join(str(dump_dir), str(int(ti)))
    os.makedirs(d, exist_ok=True)
    manifest = {}
    items = {"mesh_pos": (mesh_pos, F32), "gt": (gt, F32), "prediction": (prediction, F32), "error": (error, F32), "timesteps": (timesteps, F32)}
    if cells is not None:
        items["cells"] = (cells, np.int32)
    for name, (arr, dt) in items.items():
        a = np.ascontiguousarray(arr, dtype=dt)
        a.tofile(os.path.join(d, name + ".bin"))
        manifest[name] = {"dtype": "Int32" if dt is np.int32 else "Float32", "size": list(reversed(a.shape))}
    with open(os.path.join(d, "manifest.json"), "w") as f:
        json.dump(manifest, f)
    return d


def solver_training_euler(rhs, vjp, x0, gt, dt, val_mask, n_scale):
    """What train_step(::SolverStrategy) computes for SolverTraining with a fixed-step Euler solver (reference
    src/strategies.jl:175-196, 257-292), written as the discrete adjoint the sensitivity algorithm evaluates through VJPs of
    the right-hand side:
        x_{k+1} = x_k + dt f(x_k),   loss = mean(((n_norm(gt_k) - n_norm(x_k)) ^ 2) .* val_mask)  over k = 0..K, nodes, fields
        a_K = dL/dx_K;   a_k = dL/dx_k + a_{k+1} + dt J_x(x_k)^T a_{k+1};   gs = sum_k dt J_p(x_k)^T a_{k+1}
    rhs(x) -> f(x) [N][O];  vjp(x, lam) -> (lam^T df/dx, lam^T df/dps)  (Engine.ode_step / Engine.ode_vjp with the static
    inputs bound);  gt [K+1][N][O];  n_scale [O]: scale of the field normaliser (its shift cancels in the difference).
    Returns (gs, loss, xs)."""
    gt = np.asarray(gt, np.float64)
    K = gt.shape[0] - 1
    vm = np.asarray(val_mask, np.float64).reshape(-1, 1)
    sc = np.asarray(n_scale, np.float64).reshape(1, -1)
    xs = [np.asarray(x0, np.float64)]
    for _ in range(K):
        xs.append(xs[-1] + dt * np.asarray(rhs(xs[-1].astype(F32)), np.float64))
    count = float((K + 1) * gt.shape[1] * gt.shape[2])
    loss = float(sum((((gt[k] - xs[k]) * sc) ** 2 * vm).sum() for k in range(K + 1)) / count)

    def dl_dx(k):
        return -2.0 * sc * sc * (gt[k] - xs[k]) * vm / count

    a = dl_dx(K)
    gs = None
    for k in range(K - 1, -1, -1):
        xbar, g = vjp(xs[k].astype(F32), (dt * a).astype(F32))
        gs = np.asarray(g, np.float64) if gs is None else gs + g
        a = dl_dx(k) + a + np.asarray(xbar, np.float64)
    return gs, loss, xs
