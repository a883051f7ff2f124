"""From a dataset file to the hot path: a trajectory read from the HDF5 fixture (tests/golden/h5/grid3d, written by h5py in the layout of the
reference's docs/src/training_data.md) goes through create_base_graph / build_graph (reference src/graph.jl:25-97) into the engine, and the
model output equals the float64 oracle's on the same graph -- the `.h5` arm of the loader feeding `mgn.model(graph, ps, st)` the way
train_network / eval_network do (src/MeshGraphNets.jl:360, 596; src/solve.jl:198-200)."""
import os

import numpy as np
import pytest
import torch   # noqa: F401

import mgn_amd
import mgn_oracle as orc
from mgn_amd import hdf5_lite as h5
from mgn_amd import reference_api as ra
from util import TOL_15, rel_max

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not h5.available(), reason="libhdf5 not present on this machine")]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")


def test_h5_trajectory_through_the_graph_prologue_into_the_engine():
    ds = ra.load_dataset(os.path.join(GOLD, "grid3d"), False)
    meta = ds.meta
    traj = next(iter(ds.ch))
    N = traj["mesh_pos"].shape[1]
    # structured-grid edges as the loader made them (1-based pairs, sorted), node types 0..2, 3-D positions -> Fe = 4
    onehot, s, r, ef = ra.create_base_graph(traj, 2, 0)
    assert onehot.shape == (N, 3) and ef.shape == (s.size, 4) and s.min() >= 0 and s.max() < N
    e1 = traj["edges"]
    assert s.size == 2 * e1.shape[0] and np.array_equal(s[:e1.shape[0]], e1[:, 0] - 1) and np.array_equal(r[:e1.shape[0]], e1[:, 1] - 1)
    fields = meta["target_features"]                                   # ["velocity"], dim 2
    Fn, Fe, O, L, mps = 2 + 3, 4, 2, 128, 3
    cfg = dict(Fn=Fn, Fe=Fe, O=O, L=L, hidden_layers=2, mps=mps)
    ps = orc.init_params(Fn, Fe, O, L, 2, mps, seed=4, ln_jitter=0.1)

    class Mgn:   # the fields build_graph reads (normalisers as the reference selects them for offline statistics)
        n_norm = {"velocity": ra.NormaliserOfflineMeanStd(np.array([0.1, -0.2], np.float32), np.array([1.5, 0.7], np.float32)),
                  "node_type": ra.NormaliserOfflineMinMax(0.0, 1.0)}
        e_norm = ra.NormaliserOfflineMeanStd(ef.mean(0), ef.std(0) + 1e-3)

    eng = mgn_amd.Engine(Fn, Fe, O, L, 2, mps)
    eng.set_params(ps)
    eng.set_graph(s, r, N)
    for datapoint in (0, 2):
        g = ra.build_graph(Mgn, traj, fields, datapoint, onehot, ef, s, r)
        assert g.nf.shape == (N, Fn) and g.ef.shape == (s.size, Fe)
        out = eng.forward(g.nf, g.ef)
        assert rel_max(out, orc.forward(ps, cfg, g.nf, g.ef, s, r)) <= TOL_15
    eng.close()
