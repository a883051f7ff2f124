"""N4 data formats (CPU): the native TFRecord / tf.train.Example reader against a writer implemented here from the
published formats (TFRecord framing with masked CRC-32C; protobuf wire format of tf.train.Example), on a
cylinder_flow-shaped trajectory laid out by the reference's own meta.json schema (examples/cylinder_flow/meta.json:
shapes [T, -1, dim], static features stored once)."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

import mgn_amd
from mgn_amd import reference_api as ra
from mgn_amd import synth


# ---- writer side (test infrastructure) ----------------------------------------------------------
def crc32c_py(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ 0x82F63B78 if crc & 1 else crc >> 1
    return crc ^ 0xFFFFFFFF


def masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def ld(field, payload):   # length-delimited field
    return varint((field << 3) | 2) + varint(len(payload)) + payload


def example_bytes(features):
    """features: {name: ("bytes", b"...") | ("float", array) | ("int64", array) | ("float_unpacked", array)}"""
    entries = b""
    for name, (kind, val) in features.items():
        if kind == "bytes":
            feat = ld(1, ld(1, val))
        elif kind == "float":
            feat = ld(2, ld(1, np.asarray(val, "<f4").tobytes()))
        elif kind == "float_unpacked":
            feat = ld(2, b"".join(varint((1 << 3) | 5) + struct.pack("<f", v) for v in val))
        else:
            feat = ld(3, ld(1, b"".join(varint(int(v) & 0xFFFFFFFFFFFFFFFF) for v in val)))
        entries += ld(1, ld(1, name.encode()) + ld(2, feat))
    return ld(1, entries)


def write_tfrecord(path, records, corrupt=None):
    with open(path, "wb") as f:
        for i, rec in enumerate(records):
            head = struct.pack("<Q", len(rec))
            body = bytearray(rec)
            crc = masked(crc32c_py(rec))
            if corrupt == i:
                body[len(body) // 2] ^= 0x40
            f.write(head + struct.pack("<I", masked(crc32c_py(head))) + bytes(body) + struct.pack("<I", crc))


META = {
    "dt": 0.01, "trajectory_length": 5, "n_trajectories": 2, "dims": 2,
    "feature_names": ["cells", "mesh_pos", "node_type", "velocity"], "target_features": ["velocity"],
    "features": {
        "cells": {"type": "static", "dim": 3, "shape": [1, -1, 3], "dtype": "int32"},
        "mesh_pos": {"type": "static", "dim": 2, "shape": [1, -1, 2], "dtype": "float32"},
        "node_type": {"type": "static", "dim": 1, "shape": [1, -1, 1], "dtype": "int32", "onehot": True, "data_min": 0, "data_max": 6},
        "velocity": {"type": "dynamic", "dim": 2, "shape": [5, -1, 2], "dtype": "float32"},
        "pressure": {"type": "dynamic", "dim": 1, "shape": [5, -1, 1], "dtype": "float32"},
    },
}


def trajectory(seed, n_points):
    pos, cells, node_type, vel = synth.mesh_cyl(seed, n_points)
    rng = np.random.default_rng(seed)
    N = pos.shape[0]
    velocity = (vel[None] * (1.0 + 0.05 * rng.standard_normal((5, N, 2)))).astype(np.float32)
    pressure = rng.standard_normal((5, N, 1)).astype(np.float32)
    return dict(cells=cells.astype(np.int32), mesh_pos=pos.astype(np.float32), node_type=node_type.astype(np.int32)[:, None],
                velocity=velocity, pressure=pressure)


def test_crc32c_known_answers(lib_built):
    lib = mgn_amd.load()
    for data, want in ((b"123456789", 0xE3069283), (b"", 0), (bytes(32), 0x8A9136AA), (bytes([0xFF] * 32), 0x62A8AB43)):
        assert lib.mgn_crc32c(data, len(data)) == want
    blob = np.random.default_rng(0).integers(0, 256, 1000, dtype=np.uint8).tobytes()
    for n in (1, 7, 8, 9, 63, 1000):
        assert lib.mgn_crc32c(blob, n) == crc32c_py(blob[:n])


def test_tfrecord_roundtrip_and_parse_data(lib_built, tmp_path):
    trajs = [trajectory(1, 60), trajectory(2, 45)]          # different meshes: the -1 dimension differs per record
    recs = [example_bytes({k: ("bytes", np.ascontiguousarray(v).tobytes()) for k, v in t.items()}) for t in trajs]
    write_tfrecord(tmp_path / "train.tfrecord", recs)
    (tmp_path / "meta.json").write_text(json.dumps(META))
    meta, it = ra.load_dataset(str(tmp_path), True)
    got = list(it)
    assert len(got) == 2
    for t, g in zip(trajs, got):
        N = t["mesh_pos"].shape[0]
        assert g["velocity"].shape == (5, N, 2) and np.array_equal(g["velocity"], t["velocity"])
        assert g["pressure"].shape == (5, N, 1)
        # static features are stored once and repeated over the trajectory (src/dataset.jl:70-72)
        assert g["mesh_pos"].shape == (5, N, 2) and all(np.array_equal(g["mesh_pos"][k], t["mesh_pos"]) for k in range(5))
        assert g["cells"].dtype == np.int32 and np.array_equal(g["cells"][0], t["cells"])
        assert np.array_equal(g["node_type"][3, :, 0], t["node_type"][:, 0])
        # and they feed the graph prologue the way the reference does (src/graph.jl:25-55, first time step)
        s, r = ra.triangles_to_edges(g["cells"][0])
        assert s.size == r.size and s.max() < N


def test_float_and_int64_lists_and_unknown_key(lib_built, tmp_path):
    rec = example_bytes({"a": ("float", [1.5, -2.0, 3.25]), "b": ("int64", [1, -2, 1 << 40]), "c": ("float_unpacked", [0.5, 0.25]),
                         "raw": ("bytes", b"\x01\x02\x03")})
    write_tfrecord(tmp_path / "x.tfrecord", [rec, rec])
    rd = ra.TFRecordReader(tmp_path / "x.tfrecord")
    ex = next(rd)
    assert ex["a"][0] == 2 and np.array_equal(np.frombuffer(ex["a"][1], "<f4"), np.array([1.5, -2.0, 3.25], np.float32))
    assert ex["b"][0] == 3 and np.array_equal(np.frombuffer(ex["b"][1], "<i8"), np.array([1, -2, 1 << 40]))
    assert ex["c"][0] == 2 and np.array_equal(np.frombuffer(ex["c"][1], "<f4"), np.array([0.5, 0.25], np.float32))
    assert ex["raw"] == (1, b"\x01\x02\x03")
    with pytest.raises(KeyError):
        ra.parse_data(ex, {"features": {"missing": {"dtype": "float32", "shape": [1], "type": "dynamic"}}, "trajectory_length": 1})
    lib = mgn_amd.load()
    assert lib.mgn_tfrecord_feature(rd.h, b"missing", None, None, None) == -1
    next(rd)
    with pytest.raises(StopIteration):
        next(rd)


def test_corruption_is_detected(lib_built, tmp_path):
    rec = example_bytes({"v": ("bytes", bytes(range(200)))})
    write_tfrecord(tmp_path / "bad.tfrecord", [rec, rec, rec], corrupt=1)
    rd = ra.TFRecordReader(tmp_path / "bad.tfrecord")
    next(rd)
    with pytest.raises(ValueError, match="CRC mismatch at record 1"):
        next(rd)
    # truncated file
    data = (tmp_path / "bad.tfrecord").read_bytes()
    (tmp_path / "cut.tfrecord").write_bytes(data[:len(data) // 2 - 3])
    rd = ra.TFRecordReader(tmp_path / "cut.tfrecord", verify_crc=False)
    next(rd)
    with pytest.raises(ValueError, match="truncated"):
        next(rd)
    with pytest.raises(FileNotFoundError):
        ra.TFRecordReader(tmp_path / "nope.tfrecord")


def test_dump_rollout_layout(tmp_path):
    """N4 remainder: the raw dump julia/write_trajectories.jl converts to trajectories.h5 (reference src/MeshGraphNets.jl:638-669)
    holds the bytes and the sizes of the Julia arrays"""
    import json
    T, N, O = 3, 5, 2
    pred = np.arange(T * N * O, dtype=np.float32).reshape(T, N, O)
    d = ra.dump_rollout(tmp_path, 1, np.zeros((N, 2), np.float32), pred, pred + 1, np.ones((T, O), np.float32), np.arange(T, dtype=np.float32),
                        cells=np.zeros((4, 3), np.int32))
    m = json.load(open(os.path.join(d, "manifest.json")))
    assert m["prediction"] == {"dtype": "Float32", "size": [O, N, T]} and m["cells"]["dtype"] == "Int32" and m["cells"]["size"] == [3, 4]
    raw = np.fromfile(os.path.join(d, "prediction.bin"), np.float32)
    # Julia reads it column-major as (O x N x T): element (o, n, t) at o + O (n + N t)  ==  our [t][n][o]
    assert raw[1 + O * (2 + N * 1)] == pred[1, 2, 1] + 1

