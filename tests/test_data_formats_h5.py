"""N4 data formats (CPU): the `.h5` / `.jld2` arms of the reference's dataset loader (src/dataset.jl:118-352) and its evaluation output
`trajectories.h5` (src/MeshGraphNets.jl:638-669), on libhdf5 through `hdf5_lite`.

Fixtures under tests/golden/h5/ were written by h5py (tests/golden/make_h5_fixtures.py) in the layout the reference documents
(docs/src/training_data.md); `expected.npz` holds the arrays the files were generated from.  Files written HERE are read back by the
binding, and -- where the image has them -- by h5py and h5dump, two readers this package does not share code with."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import mgn_amd  # noqa: F401
from mgn_amd import dataset_h5 as dh
from mgn_amd import hdf5_lite as h5
from mgn_amd import reference_api as ra

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
H5PY = "/opt/conda/bin/python3.9"
pytestmark = pytest.mark.skipif(not h5.available(), reason="libhdf5 not present on this machine")


def have_h5py():
    return os.path.exists(H5PY) and subprocess.run([H5PY, "-c", "import h5py"], capture_output=True).returncode == 0


def test_binding_reads_what_h5py_wrote():
    assert h5.version() >= (1, 10, 0)
    with h5.File(os.path.join(GOLD, "grid3d", "train.h5")) as f:
        assert f.keys() == ["run_a", "run_b", "run_c"]                    # name order, whatever the creation order was
        g = f.open_group("run_b")
        assert "cl_mesh[1,1,1].pos[1]" in g and "nope" not in g
        assert len(g.keys()) == 1 + 12 * (3 + 1 + 2 + 2 + 1)
        v = g.read("cl_mesh[2,1,2].velocity[2]")
        assert v.dtype == np.float32 and v.shape == (6,)
        s = g.read("cl_mesh[3,2,2].pos[3]")
        assert s.shape == () and s.dtype == np.float32
        assert g.read("cl_mesh[1,2,1].cellType").dtype == np.int32
        assert g.read("time").dtype == np.float64
        with pytest.raises(KeyError):
            g.read("missing")
        with pytest.raises(KeyError):
            f.open_group("run_z")
    with h5.File(os.path.join(GOLD, "line1d", "test.h5")) as f:
        g = f["only"]
        e = g.read("edge_list")                                          # compound {int32, int64}
        assert e.dtype.names == ("first", "second") and e["first"].tolist() == [1, 3, 2, 5, 4, 1] and e["second"].dtype == np.int64
        u = g.read("u")                                                  # chunked, gzip + shuffle
        want = np.load(os.path.join(GOLD, "line1d", "expected.npz"))["line1d/test/only/u"]
        assert np.array_equal(u, want[:, :, 0])
    with pytest.raises(FileNotFoundError):
        h5.File(os.path.join(GOLD, "nope.h5"))
    with pytest.raises(h5.Hdf5Error):
        h5.File(os.path.join(GOLD, "grid3d", "meta.json"))               # not an HDF5 file


def test_dims_to_li_and_create_edges_known_answers():
    # LinearIndices((3, 2, 2)): column-major, 1-based (reference src/utils.jl:264-267)
    assert dh.dims_to_li([3, 2, 2], [1, 1, 1]) == 1 and dh.dims_to_li([3, 2, 2], [3, 1, 1]) == 3
    assert dh.dims_to_li([3, 2, 2], [1, 2, 1]) == 4 and dh.dims_to_li([3, 2, 2], [3, 2, 2]) == 12
    with pytest.raises(IndexError):
        dh.dims_to_li([3, 2, 2], [4, 1, 1])
    # 1-D chain (src/dataset.jl:378-381)
    assert dh.create_edges([4], np.zeros((1, 4, 1), np.int32), []) == [[1, 2], [2, 3], [3, 4]]
    with pytest.raises(ValueError, match="2D-Meshes"):
        dh.create_edges([2, 2], np.zeros((1, 4, 1), np.int32), [])
    # 2 x 1 x 2 box, worked by hand: li(x, y, z) = x + 2 (z - 1): nodes 1..4; +x edges 1-2, 3-4; +z edges 1-3, 2-4; loop order x, y, z
    nt = np.zeros((1, 4, 1), np.int32)
    assert dh.create_edges([2, 1, 2], nt, []) == [[1, 2], [1, 3], [3, 4], [2, 4]]
    # node 3 of a type without edges: its edges vanish and it gets one self-edge, where the loop meets it (src/dataset.jl:402-406)
    nt[0, 2, 0] = 5
    assert dh.create_edges([2, 1, 2], nt, [5]) == [[1, 2], [3, 3], [2, 4]]


def test_grid3d_dataset_against_the_generating_arrays():
    exp = np.load(os.path.join(GOLD, "grid3d", "expected.npz"))
    ds = ra.load_dataset(os.path.join(GOLD, "grid3d"), True)
    assert ds.meta["n_trajectories"] == 3 and ds.meta["n_trajectories_valid"] == 1
    assert ds.file.endswith("train.h5") and ds.file_valid.endswith("valid.h5")
    for (split, names, ch) in (("train", ["run_a", "run_b", "run_c"], ds.ch), ("valid", ["v1"], ds.ch_valid)):
        got = list(ch)
        assert len(got) == len(names)
        for name, t in zip(names, got):                                   # trajectories come in key (name) order
            for fn, shape, dt in (("mesh_pos", (1, 12, 3), np.float32), ("node_type", (1, 12, 1), np.int32), ("velocity", (4, 12, 2), np.float32),
                                  ("stress", (4, 12, 3), np.float32), ("stress.ev", (4, 12, 2), np.float32), ("temp", (4, 12, 1), np.float64)):
                assert t[fn].shape == shape and t[fn].dtype == dt and t[fn].flags.c_contiguous
                assert np.array_equal(t[fn], exp[f"{split}/{name}/{fn}"]), (name, fn)
            assert t["dt"].dtype == np.float32 and np.array_equal(t["dt"], exp[f"{split}/{name}/dt"])
            # edges: create_edges with the trajectory's node types, sorted as vectors (src/dataset.jl:343-346)
            want = sorted(dh.create_edges([3, 2, 2], t["node_type"], [2]))
            assert t["edges"].dtype == np.int32 and t["edges"].tolist() == want
            no_edge = set(np.nonzero(t["node_type"][0, :, 0] == 2)[0] + 1)
            for a, b in t["edges"]:
                assert (a == b and a in no_edge) or (a not in no_edge and b not in no_edge)
    test = ra.load_dataset(os.path.join(GOLD, "grid3d"), False)
    assert test.ch_valid is None and test.meta["n_trajectories"] == 2 and "n_trajectories_valid" not in test.meta
    got = list(test.ch)
    assert np.array_equal(got[0]["velocity"], exp["test/t0/velocity"]) and np.array_equal(got[1]["temp"], exp["test/t1/temp"])


def test_whole_mesh_datasets_custom_edges_and_the_jld2_arm():
    exp = np.load(os.path.join(GOLD, "line1d", "expected.npz"))
    ds = ra.load_dataset(os.path.join(GOLD, "line1d"), False)
    (t,) = list(ds.ch)
    for fn in ("mesh_pos", "node_type", "u"):
        assert np.array_equal(t[fn], exp[f"line1d/test/only/{fn}"])
    assert t["u"].shape == (3, 6, 1) and t["dt"].shape == () and t["dt"] == np.float32(0.5)
    # custom edges: the compound dataset, minus edges touching exclude_node_indices = [6], sorted as vectors
    assert t["edges"].tolist() == [[1, 2], [1, 3], [2, 3], [3, 2], [4, 5]]
    # ... and the same pairs stored as a vector of fixed arrays (H5T_ARRAY[2] elements; a rank-1 dataset that numpy shows as (E, 2))
    with h5.File(os.path.join(GOLD, "line1d", "test.h5"), "r") as f:
        g = f.open_group("only")
        assert g.rank("edge_list_arr") == 1 and g.read("edge_list_arr").shape == (6, 2)
        assert sorted(dh.read_edges(g, "edge_list_arr", None, [], [6])) == [[1, 2], [1, 3], [2, 3], [3, 2], [4, 5]]
        with pytest.raises(TypeError, match="2-dimensional"):
            dh.read_edges(g, "u", None, [], [])                 # a matrix is not a vector of pairs (Julia: MethodError in filter!)
    # train.jld2 is preferred over train.h5 (src/dataset.jl:94-100); no custom edges there: the 1-D chain
    jl = ra.load_dataset(os.path.join(GOLD, "line1d_jld"), True)
    assert jl.file.endswith("train.jld2") and jl.file_valid.endswith("valid.jld2")
    assert jl.meta["n_trajectories"] == 2 and jl.meta["n_trajectories_valid"] == 1
    a, b = list(jl.ch)
    assert np.array_equal(a["u"], exp["line1d_jld/train/a/u"]) and np.array_equal(b["node_type"], exp["line1d_jld/train/b/node_type"])
    assert a["edges"].tolist() == [[i, i + 1] for i in range(1, 6)]
    (v,) = list(jl.ch_valid)
    assert np.array_equal(v["u"], exp["line1d_jld/valid/v/u"])
    # "Custom edge definition is not supported for JLD2 files." (src/dataset.jl:326-328)
    meta = json.load(open(os.path.join(GOLD, "line1d", "meta.json")))
    with pytest.raises(ValueError, match="not supported for JLD2"):
        dh.read_trajectory(os.path.join(GOLD, "line1d_jld", "train.jld2"), "a", meta, is_jld=True)


def test_layouts_the_reference_refuses_are_refused(tmp_path):
    """Julia's shape rules, restated: a static whole-mesh VECTOR of 6 does not broadcast into 1 x 6 x 1, a dynamic dataset shorter than
    trajectory_length is a BoundsError, a missing key leaves zeros (the reference never checks that a feature matched anything)."""
    meta = json.load(open(os.path.join(GOLD, "line1d", "meta.json")))
    del meta["custom_edges"]
    p = str(tmp_path / "t.h5")
    with h5.File(p, "w") as f:
        with f.create_group("bad_static") as g:
            g["x"] = np.zeros(6, np.float32)
            g["kind"] = np.zeros((6, 1), np.int32)
            g["u"] = np.zeros((3, 6), np.float32)
            g["t"] = np.float32(1)
        with f.create_group("short") as g:
            g["x"] = np.zeros((6, 1), np.float32)
            g["kind"] = np.zeros((6, 1), np.int32)
            g["u"] = np.zeros((2, 6), np.float32)
            g["t"] = np.float32(1)
        with f.create_group("no_u") as g:
            g["x"] = np.ones((6, 1), np.float32)
            g["kind"] = np.zeros((6, 1), np.int32)
            g["t"] = np.arange(3, dtype=np.float64)
    with pytest.raises(dh.DimensionMismatch):
        dh.read_trajectory(p, "bad_static", meta)
    with pytest.raises(IndexError):
        dh.read_trajectory(p, "short", meta)
    t = dh.read_trajectory(p, "no_u", meta)
    assert not t["u"].any() and t["mesh_pos"].all() and t["dt"].dtype == np.float32 and t["dt"].tolist() == [0, 1, 2]
    meta["features"]["u"]["type"] = "sometimes"
    with pytest.raises(ValueError, match="static or dynamic"):
        dh.read_trajectory(p, "no_u", meta)
    with pytest.raises(FileNotFoundError):
        ra.load_dataset(str(tmp_path), True)                              # no train.tfrecord / .jld2 / .h5


def test_trajectories_h5_schema_and_independent_readers(tmp_path):
    rng = np.random.default_rng(0)
    T, N, O = 4, 7, 2
    trajs = {}
    for ti in (1, 3):                                                    # counter 2 has no entry: its (empty) group still exists
        pred = rng.standard_normal((T, N, O)).astype(np.float32)
        trajs[ti] = {"mesh_pos": rng.standard_normal((N, 2)).astype(np.float32), "gt": pred + 1, "prediction": pred,
                     "error": rng.random((T, O)).astype(np.float32), "timesteps": np.arange(T, dtype=np.float32) * 0.01,
                     "cells": rng.integers(0, N, (5, 3)).astype(np.int32)}
    path = dh.write_trajectories_h5(str(tmp_path / "euler"), trajs)
    assert path.endswith(os.path.join("euler", "trajectories.h5"))
    with h5.File(path) as f:
        assert f.keys() == ["1", "2", "3"] and f.open_group("2").keys() == []
        g = f.open_group("3").open_group("prediction")
        assert g.keys() == ["data", "size"]
        assert g.read("size").dtype == np.int64 and g.read("size").tolist() == [O, N, T]          # the Julia size
        assert g.read("data").shape == (T * N * O,)                                              # reshape(value, length(value))
        # column-major flattening of the Julia O x N x T array == the bytes of [T][N][O]
        jl = trajs[3]["prediction"].transpose(2, 1, 0)
        assert np.array_equal(g.read("data"), jl.reshape(-1, order="F"))
        assert f["1"]["cells"].read("data").dtype == np.int32
    back = dh.read_trajectories_h5(path)
    assert set(back) == {1, 2, 3} and back[2] == {}
    for ti in (1, 3):
        for k, v in trajs[ti].items():
            assert np.array_equal(back[ti][k], v) and back[ti][k].dtype == v.dtype
    if have_h5py():
        code = ("import h5py, numpy as np, sys\n"
                "f = h5py.File(sys.argv[1], 'r')\n"
                "assert sorted(f) == ['1', '2', '3']\n"
                "d = f['3/prediction/data'][...]; s = f['3/prediction/size'][...]\n"
                "assert d.dtype == np.float32 and s.dtype == np.int64\n"
                "np.save(sys.argv[2], d.reshape(s[::-1]))\n")
        out = str(tmp_path / "pred.npy")
        subprocess.run([H5PY, "-c", code, path, out], check=True)
        assert np.array_equal(np.load(out), trajs[3]["prediction"])
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        txt = subprocess.run([h5dump, "-H", path], check=True, capture_output=True, text=True).stdout
        assert 'GROUP "timesteps"' in txt and "H5T_IEEE_F32LE" in txt and "H5T_STD_I64LE" in txt and "H5T_STD_I32LE" in txt


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(h5, "_lib", None)
    monkeypatch.setenv("MGN_HDF5_LIB", "/nonexistent/libhdf5.so")
    with pytest.raises(h5.Hdf5Unavailable):
        h5.File(os.path.join(GOLD, "line1d", "test.h5"))
