"""Writes the HDF5 dataset fixtures under tests/golden/h5/ with h5py -- an implementation independent of this package's libhdf5 binding
(run with an interpreter that has h5py: `/opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py` in the build image).

Layout as docs/src/training_data.md of the reference prescribes it (one group per trajectory, one dataset per key; keys
`name[%d,...]` per mesh point, a trailing `[c]` for split features) and as src/dataset.jl:194-352 reads it.  The expected arrays
(`expected.npz`, this package's [time][count][dim] order) are the arrays the files were generated FROM, so the reader is checked against
the data, not against itself.

    grid3d/   dims [3, 2, 2]: per-point keys, a split static feature, a split dynamic feature stored longer than trajectory_length
              (two-dimensional per-point datasets dim x T as well), has_ev, node types that switch edges off, train / valid / test
    line1d/   dims [6]: whole-mesh datasets (no mesh index in the key), gzip + shuffle chunks, custom edges as a compound dataset,
              exclude_node_indices; stored as train.jld2 / valid.jld2 (JLD2 files are HDF5 files) and test.h5"""
import json
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5")


def li(dims, idx):   # 0-based column-major linear index of a 1-based multi-index
    out, stride = 0, 1
    for i, d in zip(idx, dims):
        out += (i - 1) * stride
        stride *= d
    return out


def grid3d():
    d = os.path.join(HERE, "grid3d")
    os.makedirs(d, exist_ok=True)
    dims, T, stored = [3, 2, 2], 4, 6
    count = int(np.prod(dims))
    meta = {
        "dt": "time", "trajectory_length": T, "dims": dims, "no_edges_node_types": [2],
        "feature_names": ["mesh_pos", "node_type", "velocity", "stress", "temp"], "target_features": ["velocity"],
        "features": {
            "mesh_pos": {"key": "cl_mesh[%d,%d,%d].pos", "split": True, "dim": 3, "type": "static", "dtype": "float32"},
            "node_type": {"key": "cl_mesh[%d,%d,%d].cellType", "dim": 1, "type": "static", "dtype": "int32", "onehot": True,
                          "data_min": 0, "data_max": 3},
            "velocity": {"key": "cl_mesh[%d,%d,%d].velocity", "split": True, "dim": 2, "type": "dynamic", "dtype": "float32"},
            "stress": {"key": "cl_mesh[%d,%d,%d].stress", "dim": 3, "type": "dynamic", "dtype": "float32", "has_ev": True},
            "temp": {"key": "cl_mesh[%d,%d,%d].temp", "type": "dynamic", "dtype": "float64"},
        },
    }
    with open(os.path.join(d, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    expected = {}
    for fname, names, seed in (("train.h5", ["run_b", "run_a", "run_c"], 1), ("valid.h5", ["v1"], 2), ("test.h5", ["t1", "t0"], 3)):
        rng = np.random.default_rng(seed)
        with h5py.File(os.path.join(d, fname), "w") as f:
            for name in names:
                g = f.create_group(name)
                pos = rng.standard_normal((count, 3)).astype(np.float32)
                ntype = rng.integers(0, 3, count).astype(np.int32)
                vel = rng.standard_normal((stored, count, 2)).astype(np.float32)
                stress = rng.standard_normal((stored, count, 3)).astype(np.float32)
                ev = rng.standard_normal((stored, count, 2)).astype(np.float32)
                temp = rng.standard_normal((stored, count))
                time = (0.01 * np.arange(stored)).astype(np.float64)
                g["time"] = time
                for x in range(1, dims[0] + 1):
                    for y in range(1, dims[1] + 1):
                        for z in range(1, dims[2] + 1):
                            n = li(dims, (x, y, z))
                            key = f"cl_mesh[{x},{y},{z}]"
                            for c in range(3):
                                g[f"{key}.pos[{c + 1}]"] = pos[n, c]                       # scalar datasets
                            g[f"{key}.cellType"] = ntype[n]
                            for c in range(2):
                                g[f"{key}.velocity[{c + 1}]"] = vel[:, n, c]               # 1-D, longer than trajectory_length
                            # two-dimensional per-point datasets: Julia sees dim x stored, HDF5 stores (stored, dim)
                            g[f"{key}.stress"] = stress[:, n, :]
                            g[f"{key}.stress.ev"] = ev[:, n, :]
                            g[f"{key}.temp"] = temp[:, n]
                tag = fname.split(".")[0] + "/" + name
                expected[tag + "/mesh_pos"] = pos[None]
                expected[tag + "/node_type"] = ntype[None, :, None]
                expected[tag + "/velocity"] = vel[:T]
                expected[tag + "/stress"] = stress[:T]
                expected[tag + "/stress.ev"] = ev[:T]
                expected[tag + "/temp"] = temp[:T, :, None]
                expected[tag + "/dt"] = time.astype(np.float32)
    np.savez_compressed(os.path.join(d, "expected.npz"), **expected)


def line1d():
    d = os.path.join(HERE, "line1d")
    os.makedirs(d, exist_ok=True)
    dims, T, stored = [6], 3, 3
    meta = {
        "dt": "t", "trajectory_length": T, "dims": dims, "custom_edges": "edge_list", "exclude_node_indices": [6],
        "feature_names": ["mesh_pos", "node_type", "u"], "target_features": ["u"],
        "features": {
            "mesh_pos": {"key": "x", "dim": 1, "type": "static", "dtype": "float32"},
            "node_type": {"key": "kind", "dim": 1, "type": "static", "dtype": "int32"},
            "u": {"key": "u", "dim": 1, "type": "dynamic", "dtype": "float32"},
        },
    }
    meta_jld = {k: v for k, v in meta.items() if k not in ("custom_edges", "exclude_node_indices")}
    for sub, m in (("", meta), ("_jld", meta_jld)):
        os.makedirs(d + sub, exist_ok=True)
        with open(os.path.join(d + sub, "meta.json"), "w") as f:
            json.dump(m, f, indent=1)
    expected = {}
    edge_t = np.dtype([("first", "<i4"), ("second", "<i8")])
    for folder, fname, names, seed in ((d, "test.h5", ["only"], 5), (d + "_jld", "train.jld2", ["a", "b"], 6),
                                       (d + "_jld", "valid.jld2", ["v"], 7)):
        rng = np.random.default_rng(seed)
        with h5py.File(os.path.join(folder, fname), "w") as f:
            for name in names:
                g = f.create_group(name)
                x = np.linspace(0, 1, 6).astype(np.float32)
                kind = rng.integers(0, 2, 6).astype(np.int32)
                u = rng.standard_normal((stored, 6)).astype(np.float32)
                g["x"] = x.reshape(6, 1)                                   # Julia: 1 x 6 (what `A[:, :, :] .= data` needs for dim x 6 x 1)
                g["kind"] = kind.reshape(6, 1)
                g.create_dataset("u", data=u, chunks=(1, 3), compression="gzip", shuffle=True)   # Julia: 6 x T
                g["t"] = np.float32(0.5)
                edges = np.array([(1, 2), (3, 2), (2, 3), (5, 6), (4, 5), (1, 3)], dtype=edge_t)
                g["edge_list"] = edges
                # the same pairs as a vector of fixed arrays (H5T_ARRAY[2] elements): read_edges' other documented element kind
                arr = g.create_dataset("edge_list_arr", (6,), dtype=np.dtype(("<i4", (2,))))
                arr[...] = np.array([(1, 2), (3, 2), (2, 3), (5, 6), (4, 5), (1, 3)], np.int32)
                tag = os.path.basename(folder) + "/" + fname.split(".")[0] + "/" + name
                expected[tag + "/mesh_pos"] = x[None, :, None]
                expected[tag + "/node_type"] = kind[None, :, None]
                expected[tag + "/u"] = u[:, :, None]
                expected[tag + "/dt"] = np.float32(0.5)
    np.savez_compressed(os.path.join(d, "expected.npz"), **expected)


if __name__ == "__main__":
    grid3d()
    line1d()
    print("written under", HERE)
