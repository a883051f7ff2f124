"""bench.py's `secondary.scaling_model` (2 / 4 / 8 GPUs predicted from one: the real partition's shares through the staged schedule) on a
small mesh: it has to run on a one-GPU box -- the driver's bench run is not the place to find out that it does not."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mgn_amd  # noqa: E402

pytestmark = pytest.mark.gpu


def test_scaling_model_runs_and_adds_up():
    import torch
    import bench
    pos, s, r = mgn_amd.synth.mesh_1m(1234, 180, 180)
    N, E = pos.shape[0], int(s.size)
    ps = bench.glorot_params()

    def sync():
        torch.cuda.synchronize()

    m = bench.scaling_model(ps, pos, s, r, N, E, 1e-4, 0, sync)
    for P in (2, 4, 8):
        d = m[f"{P}_gpus"]
        assert abs(d["n_own"] - N / P) <= 1 and d["n_halo"] > 0 and d["send_rows"] > 0
        assert d["edge_tiles_boundary"] > 0 and d["edge_tiles_interior"] > d["edge_tiles_boundary"]
        assert np.isfinite(d["share_ms_per_step_staged"]) and d["share_ms_per_step_staged"] > 0
        assert d["predicted_ms_per_step"] >= d["share_ms_per_step_staged"]
        assert d["predicted_edges_per_s"] == pytest.approx(E / (d["predicted_ms_per_step"] * 1e-3))
        assert 0 < d["pack_us"] < 1e4 and 0 < d["unpack_us"] < 1e4
    assert m["8_gpus"]["share_ms_per_step_staged"] < m["2_gpus"]["share_ms_per_step_staged"]
