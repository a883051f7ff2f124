"""The `.h5` / `.jld2` branches of the reference's dataset loader and its evaluation writer, on libhdf5 (`hdf5_lite`).

    read_h5                 reference src/dataset.jl:194-352   (`read_h5!`)
    create_edges            reference src/dataset.jl:366-416
    read_edges              reference src/dataset.jl:434-450
    dims_to_li              reference src/utils.jl:264-267
    load_dataset_h5         reference src/dataset.jl:118-166   (the `jld2` / `h5` arm of `load_dataset`)
    write_trajectories_h5   reference src/MeshGraphNets.jl:638-669   (`trajectories.h5`: `/<ti>/<name>/{data,size}`)

The reference fills Julia arrays `dim x prod(dims) x T` with Julia's indexing rules (`A[coord, li, :] = data[coord, 1:T]`,
`A[coord, :, :] .= data`).  Those rules are restated here on column-major views (`_assign`, `_broadcast_assign`: shapes compared with
singleton dimensions dropped; broadcasting aligns LEADING dimensions), so a file laid out for the reference loads to the same numbers
and a file the reference would refuse with a `DimensionMismatch` is refused here.  Arrays are returned in this package's
[time][count][dim] order -- the bytes of the Julia arrays.  Indices in `edges` stay 1-based, as the reference returns them."""
from __future__ import annotations

import json
import os
import re

import numpy as np

from . import hdf5_lite as h5

_JULIA_TYPES = {"Int32": np.int32, "Int64": np.int64, "Float32": np.float32, "Float64": np.float64, "Bool": np.bool_,
                "Int8": np.int8, "Int16": np.int16, "UInt8": np.uint8}


class DimensionMismatch(ValueError):
    pass


def _julia_type(name):
    """getfield(Base, Symbol(uppercasefirst(dtype))) (src/dataset.jl:216-218): "float32" -> Float32, "int32" -> Int32, "Bool" -> Bool."""
    key = name[:1].upper() + name[1:]
    if key not in _JULIA_TYPES:
        raise ValueError(f"UndefVarError: {key} not defined")
    return _JULIA_TYPES[key]


def dims_to_li(dims, idxs):
    """LinearIndices(Tuple(dims))[idxs...] (reference src/utils.jl:264-267): 1-based column-major linear index."""
    if len(idxs) != len(dims):
        raise IndexError(f"BoundsError: {len(idxs)} indices for a mesh of {len(dims)} dimensions")
    li, stride = 0, 1
    for i, d in zip(idxs, dims):
        if not 1 <= i <= d:
            raise IndexError(f"BoundsError: index {list(idxs)} outside dims {list(dims)}")
        li += (i - 1) * stride
        stride *= d
    return li + 1


def _jl(a):
    """The Julia view (dimensions reversed, column-major) of a C-order array read from the file."""
    return np.asarray(a).T


def _nonsingleton(shape):
    return tuple(s for s in shape if s != 1)


def _assign(dest, src):
    """`A[I...] = B` for an array B (Base.setindex_shape_check): the shapes must agree once singleton dimensions are dropped;
    elements are copied in column-major order."""
    src = np.asarray(src)
    if _nonsingleton(dest.shape) != _nonsingleton(src.shape):
        raise DimensionMismatch(f"DimensionMismatch: tried to assign {src.shape} array to {dest.shape} destination")
    dest[...] = src.reshape(dest.shape, order="F")


def _broadcast_assign(dest, src):
    """`A[I...] .= B`: Julia broadcasting aligns the LEADING dimensions; a dimension of B must equal A's or be 1; extra trailing
    dimensions of B must be 1."""
    src = np.asarray(src)
    shape = list(src.shape)
    while len(shape) > dest.ndim and shape[-1] == 1:
        shape.pop()
    if len(shape) > dest.ndim:
        raise DimensionMismatch(f"DimensionMismatch: cannot broadcast {src.shape} into {dest.shape}")
    shape += [1] * (dest.ndim - len(shape))
    for s, d in zip(shape, dest.shape):
        if s != d and s != 1:
            raise DimensionMismatch(f"DimensionMismatch: cannot broadcast {src.shape} into {dest.shape}")
    dest[...] = src.reshape(shape, order="F")


def _first_t(data, coord, tl):
    """`data[coord, 1:tl]` for a two-dimensional array, `data[1:tl]` (linear indexing) otherwise (src/dataset.jl:283-289, 299-305)."""
    if data.ndim == 2:
        if data.shape[1] < tl:
            raise IndexError(f"BoundsError: {data.shape[1]} time steps stored, trajectory_length is {tl}")
        return data[:, :tl] if coord is None else data[np.asarray(coord) - 1, :tl]
    flat = data.reshape(-1, order="F")
    if flat.size < tl:
        raise IndexError(f"BoundsError: {flat.size} values stored, trajectory_length is {tl}")
    return flat[:tl]


def _key_regex(key, split):
    """src/dataset.jl:225-235: brackets escaped, `%d` -> `\\d+`, `\\[\\d+\\]` appended for split features.  Not anchored: `match`
    finds the first occurrence anywhere in a link name and the MATCHED TEXT is what is read afterwards."""
    rx = key.replace("[", "\\[").replace("]", "\\]").replace("%d", "\\d+")
    return re.compile(rx + ("\\[\\d+\\]" if split else ""))


def _bracket(m, piece):
    """split(split(m, r"(\\[|\\])")[piece], ","): Julia's split drops the delimiters; piece is 1-based."""
    parts = re.split(r"\[|\]", m)
    return [int(x) for x in parts[piece - 1].split(",")]


def create_edges(dims, node_type, no_edges_node_types):
    """create_edges(dims, node_type, no_edges_node_types) (reference src/dataset.jl:366-416).  node_type: [1][count][1] (or any array
    whose first `count` values in memory order are the types).  Returns a list of [a, b] pairs, 1-based, in the reference's order."""
    dims = [int(d) for d in dims]
    edges = []
    if len(dims) == 1:
        return [[i, i + 1] for i in range(1, dims[0])]
    if len(dims) == 2:
        raise ValueError("ArgumentError: 2D-Meshes are not supported yet")
    if len(dims) != 3:
        return edges
    nt = np.asarray(node_type)
    nt = nt[0, :, 0] if nt.ndim == 3 else nt.reshape(-1)    # node_type[1, li, 1] of the reference's (dim, count, tl) array = [tl][count][dim] here
    excluded = set(int(x) for x in no_edges_node_types)
    dx, dy, dz = dims
    li = lambda x, y, z: dims_to_li(dims, (x, y, z))       # noqa: E731
    seen_self = set()
    for x in range(1, dx + 1):
        for y in range(1, dy + 1):
            for z in range(1, dz + 1):
                a = li(x, y, z)
                if int(nt[a - 1]) not in excluded:
                    for cond, (sx, sy, sz) in ((x != dx, (1, 0, 0)), (y != dy, (0, 1, 0)), (z != dz, (0, 0, 1))):
                        if cond:
                            b = li(x + sx, y + sy, z + sz)
                            if int(nt[b - 1]) not in excluded:
                                edges.append([a, b])
                elif a not in seen_self:
                    seen_self.add(a)
                    edges.append([a, a])
    return edges


def read_edges(traj, edge_key, node_type, no_edges_node_types, exclude_node_indices):
    """read_edges(traj::Group, edge_key, ...) (reference src/dataset.jl:434-450): a one-dimensional dataset whose elements index as
    `edge[1]`, `edge[2]` -- a compound of two integers or a fixed array of two.  Edges touching an excluded node are dropped.
    `findall(x -> x in no_edges_node_types, node_type)` on the three-dimensional node_type array yields CartesianIndex values, which never
    equal an integer endpoint: as in the reference, only `exclude_node_indices` removes edges."""
    if edge_key not in traj:
        raise KeyError(f"Key '{edge_key}' not found in trajectory group '{traj.name}'")
    raw = traj.read(edge_key)
    rank = traj.rank(edge_key)
    if rank != 1:
        raise TypeError(f"MethodError: filter! on a {rank}-dimensional edge dataset (the reference needs a vector of pairs)")
    if raw.dtype.names:
        if len(raw.dtype.names) < 2:
            raise IndexError("BoundsError: edge elements need two fields")
        a, b = raw[raw.dtype.names[0]], raw[raw.dtype.names[1]]
    elif raw.ndim == 2:                       # a vector of fixed arrays (H5T_ARRAY elements): numpy shows the element's axis
        if raw.shape[1] < 2:
            raise IndexError("BoundsError: edge elements need two entries")
        a, b = raw[:, 0], raw[:, 1]
    else:
        raise IndexError("BoundsError: attempt to access a scalar edge element at index [2]")
    del node_type, no_edges_node_types
    excl = set(int(i) for i in exclude_node_indices)
    return [[int(p), int(q)] for p, q in zip(a, b) if int(p) not in excl and int(q) not in excl]


def _sorted_edges(edges):
    """hcat(sort(edges)...) (src/dataset.jl:346): vectors sorted lexicographically; returned as [E][2] (the bytes of Julia's 2 x E)."""
    if not edges:
        return np.zeros((0, 2), np.int32)
    return np.asarray(sorted(edges), dtype=np.int32).reshape(-1, 2)


def read_trajectory(datafile, k, meta, is_jld=False):
    """One iteration of `get_traj` (reference src/dataset.jl:203-349): the trajectory group `k` of `datafile` as a dictionary
    {feature: [tl][prod(dims)][dim], feature.ev: [tl][prod(dims)][2], "dt": float32 array, "edges": [E][2] int32 (1-based)}."""
    feature_names, dims, tl_all = meta["feature_names"], [int(d) for d in meta["dims"]], int(meta["trajectory_length"])
    count = int(np.prod(dims))
    out = {}
    with h5.File(datafile, "r") as file:
        traj = file.open_group(k)
        names = traj.keys()
        for fn in feature_names:
            fm = meta["features"][fn]
            dim = int(fm.get("dim", 1))
            if fm["type"] == "static":
                tl = 1
            elif fm["type"] == "dynamic":
                tl = tl_all
            else:
                raise ValueError("ArgumentError: feature type must be static or dynamic")
            dt = _julia_type(fm["dtype"])
            arrays = {fn: np.zeros((dim, count, tl), dtype=dt, order="F")}
            has_ev = bool(fm.get("has_ev", False))
            if has_ev:
                arrays[fn + ".ev"] = np.zeros((2, count, tl), dtype=dt, order="F")
            split = bool(fm.get("split", False))
            rx = _key_regex(fm["key"], split)
            matches = []
            for name in names:
                mo = rx.search(name)
                if mo is not None and mo.group(0) not in matches:
                    matches.append(mo.group(0))
            match_data = {}
            for m in matches:
                match_data[m] = _jl(traj.read(m))
                if has_ev:
                    match_data[m + ".ev"] = _jl(traj.read(m + ".ev"))
            for m, data in match_data.items():
                fn_k = fn + ".ev" if ".ev" in m else fn
                A = arrays[fn_k]
                if "]" not in m[:-1]:                                  # no mesh index in the key: the dataset covers the mesh
                    coord = _bracket(m, 2) if split else None
                    dest = A if coord is None else A[np.asarray(coord) - 1, :, :]
                    li = None
                else:
                    idx = _bracket(m, 2)
                    coord = _bracket(m, 4) if split else None
                    li = dims_to_li(dims, idx) - 1
                    dest = A[:, li, :] if coord is None else A[np.asarray(coord) - 1, li, :]
                tmp = np.zeros(dest.shape, dtype=A.dtype, order="F")
                if fm["type"] == "dynamic":
                    _assign(tmp, _first_t(data, coord, tl))
                else:
                    _broadcast_assign(tmp, data)
                if li is None:
                    if coord is None:
                        A[...] = tmp
                    else:
                        A[np.asarray(coord) - 1, :, :] = tmp
                elif coord is None:
                    A[:, li, :] = tmp
                else:
                    A[np.asarray(coord) - 1, li, :] = tmp
            for name, A in arrays.items():
                out[name] = np.ascontiguousarray(A.T)                  # [tl][count][dim]: the bytes of the Julia array
        out["dt"] = np.asarray(_jl(traj.read(meta["dt"])), dtype=np.float32)
        if "custom_edges" in meta:
            if is_jld:
                raise ValueError("ArgumentError: Custom edge definition is not supported for JLD2 files.")
            edges = read_edges(traj, meta["custom_edges"], out["node_type"], meta.get("no_edges_node_types", []),
                               meta.get("exclude_node_indices", []))
        else:
            edges = create_edges(dims, out["node_type"], meta.get("no_edges_node_types", []))
        out["edges"] = _sorted_edges(edges)
    return out


def read_h5(datafile, data_keys, meta, is_jld=False):
    """read_h5!(datafile, data_keys, meta, is_jld) (reference src/dataset.jl:194-352): a generator standing in for the Channel."""
    for k in data_keys:
        yield read_trajectory(datafile, k, meta, is_jld)


def load_dataset_h5(path, is_training, file):
    """The `jld2` / `h5` arm of load_dataset (reference src/dataset.jl:118-166).  Returns (meta, generator over the training or test
    trajectories, generator over the validation trajectories or None); meta gains n_trajectories (and n_trajectories_valid).
    Order of the trajectories: `keys(file)` -- HDF5.jl lists an .h5 file's links in name order, as here; JLD2.jl's `keys` of a .jld2 file
    returns them in the order they were written, which libhdf5 cannot see unless the file tracks creation order, so for .jld2 files the
    SET of trajectories is the reference's and their order may differ (name order here)."""
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    is_jld = file.endswith("jld2")
    with h5.File(os.path.join(path, file), "r") as df:
        data_keys = df.keys()
    meta["n_trajectories"] = len(data_keys)
    ch_valid = None
    if is_training:
        valid = os.path.join(path, "valid.jld2" if is_jld else "valid.h5")
        with h5.File(valid, "r") as fv:
            keys_valid = fv.keys()
        meta["n_trajectories_valid"] = len(keys_valid)
        ch_valid = read_h5(valid, keys_valid, meta, is_jld)
    return meta, read_h5(os.path.join(path, file), data_keys, meta, is_jld), ch_valid


# ---- evaluation output ------------------------------------------------------------------------------------------------------
def write_trajectories_h5(eval_path, trajectories):
    """`trajectories.h5` of eval_network! (reference src/MeshGraphNets.jl:638-669): one group per trajectory counter `ti` (1-based), in it
    one group per quantity with `data` = the array flattened in Julia's (column-major) order and `size` = its Julia size (Int64).
    trajectories: {ti: {"mesh_pos" | "gt" | "prediction" | "error" | "timesteps" | "cells": array}}; arrays arrive in this package's order
    ([time][count][feat], [count][feat], [time][feat], [time]) -- their C-order bytes ARE the Julia column-major bytes of
    feat x count x time etc., so `data` is the plain flattening and `size` the reversed shape.  Groups `1..max(ti)` are created even where a
    counter has no entry, as the reference's `for i in 1:maximum(...)` does."""
    os.makedirs(eval_path, exist_ok=True)
    target = os.path.join(eval_path, "trajectories.h5")
    with h5.File(target, "w") as f:
        top = max(int(t) for t in trajectories) if trajectories else 0
        groups = {i: f.create_group(str(i)) for i in range(1, top + 1)}
        try:
            for ti, items in trajectories.items():
                g = groups[int(ti)]
                for name, value in items.items():
                    a = np.ascontiguousarray(value)
                    if a.dtype == np.float64 and name != "timesteps":
                        a = a.astype(np.float32)
                    with g.create_group(name) as sub:
                        sub["data"] = a.reshape(-1)
                        sub["size"] = np.asarray(list(reversed(a.shape)), dtype=np.int64)
        finally:
            for g in groups.values():
                g.close()
    return target


def read_trajectories_h5(path):
    """Inverse of write_trajectories_h5 (tests, post-processing): {ti: {name: array in this package's order}}."""
    out = {}
    with h5.File(path, "r") as f:
        for ti in f.keys():
            g = f.open_group(ti)
            out[int(ti)] = {}
            for name in g.keys():
                sub = g.open_group(name)
                size = [int(s) for s in sub.read("size")]
                out[int(ti)][name] = sub.read("data").reshape(list(reversed(size)))
    return out
