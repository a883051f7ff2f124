"""The checkpoint surface (`load` / `save!`, reference src/MeshGraphNets.jl:282-285, 324-325, 460-471, 537-540) round-trips the state the
unchanged call sites assume: parameters, loss log, normaliser statistics, optimiser state.  CPU part: the Python twin's files and the
Julia shim's text; GPU part (tests/test_gpu_checkpoint.py): train -> save -> load -> the same right-hand side."""
import os
import re

import numpy as np
import pytest

from mgn_amd import checkpoint as ck
from mgn_amd import reference_api as ra

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norms(rng, accumulate=True):
    e = ra.NormaliserOnline(3)
    n = {"velocity": ra.NormaliserOnline(2), "node_type": ra.NormaliserOfflineMinMax(0.0, 1.0)}
    o = {"velocity": ra.NormaliserOnline(2)}
    if accumulate:
        for _ in range(3):
            e(rng.normal(2.0, 3.0, (50, 3)).astype(np.float32))
            n["velocity"](rng.normal(-1.0, 0.5, (40, 2)).astype(np.float32))
            o["velocity"](rng.normal(0.1, 0.01, (40, 2)).astype(np.float32))
    return e, n, o


def test_checkpoint_round_trips_all_four(tmp_path):
    rng = np.random.default_rng(0)
    e, n, o = _norms(rng)
    ps = rng.normal(size=1000).astype(np.float32)
    opt = ck.Adam(1e-3)
    st = opt.setup(ps)
    for _ in range(3):
        st, ps = opt.update(st, ps, rng.normal(size=ps.size).astype(np.float32))
    tr, va = ck.LossLog(), ck.LossLog()
    tr.step += [100, 200]
    tr.loss += [np.float32(0.5), np.float32(0.25)]
    va.step += [200]
    va.loss += [np.float32(0.125)]
    ck.write_checkpoint(str(tmp_path), ps, e, n, o, st, tr, va)
    assert sorted(os.listdir(tmp_path)) == sorted([ck.CKPT_PARAMS, ck.CKPT_LOG, ck.CKPT_NORMS, ck.CKPT_OPT, ck.CKPT_MANIFEST])   # no .tmp left behind

    e2, n2, o2 = _norms(rng, accumulate=False)                     # what calc_norms hands to load: empty statistics
    x = rng.normal(2.0, 3.0, (7, 3)).astype(np.float32)
    assert not np.allclose(e2(x, accumulate=False), e(x, accumulate=False))
    ps2, e2, n2, o2, st2, tr2, va2 = ck.read_checkpoint(str(tmp_path), ps.size, e2, n2, o2)
    assert np.array_equal(ps2, ps)
    for a, b in ((e, e2), (n["velocity"], n2["velocity"]), (o["velocity"], o2["velocity"])):
        assert np.array_equal(a.acc_sum, b.acc_sum) and np.array_equal(a.acc_sum_squared, b.acc_sum_squared)
        assert a.acc_count == b.acc_count and a.num_accumulations == b.num_accumulations and a.max_acc == b.max_acc
    assert np.array_equal(e2(x, accumulate=False), e(x, accumulate=False))
    y = rng.normal(size=(5, 2)).astype(np.float32)
    assert np.array_equal(ra.inverse_data(o2["velocity"], y), ra.inverse_data(o["velocity"], y))
    assert set(st2) == set(st) and all(np.array_equal(st2[k], st[k]) for k in st)
    assert tr2.step == [100, 200] and va2.step == [200] and tr2.loss == tr.loss and va2.loss == va.loss
    # the resumed optimiser continues exactly where the saved one would have
    g = rng.normal(size=ps.size).astype(np.float32)
    assert np.array_equal(opt.update(st2, ps2, g)[1], opt.update(st, ps, g)[1])
    # eval_network passes opt = nothing (src/MeshGraphNets.jl:537-540): no optimiser state is read
    assert ck.read_checkpoint(str(tmp_path), ps.size, *_norms(rng, False), want_opt_state=False)[4] is None


def test_checkpoint_refuses_what_would_evaluate_wrongly(tmp_path):
    rng = np.random.default_rng(1)
    e, n, o = _norms(rng)
    ps = np.zeros(10, np.float32)
    assert ck.read_checkpoint(str(tmp_path), 10, e, n, o) is None                      # nothing there: a fresh run
    ck.write_checkpoint(str(tmp_path), ps, e, n, o, None, ck.LossLog(), ck.LossLog())
    assert ck.read_checkpoint(str(tmp_path), 10, *_norms(rng, False))[4] is None       # opt_state None stays None
    with pytest.raises(ValueError, match="bytes"):
        ck.read_checkpoint(str(tmp_path), 11, *_norms(rng, False))                     # another model's parameters
    with pytest.raises(ValueError, match="NormaliserOnline"):                          # kinds must match the training run's
        ck.read_checkpoint(str(tmp_path), 10, ra.NormaliserOfflineMeanStd(0.0, 1.0), *_norms(rng, False)[1:])
    os.remove(tmp_path / ck.CKPT_NORMS)                                                # a checkpoint from before normalisers were stored
    os.remove(tmp_path / ck.CKPT_MANIFEST)                                             # (... and before the manifest)
    with pytest.raises(ValueError, match="online"):
        ck.read_checkpoint(str(tmp_path), 10, *_norms(rng, False))
    off = ra.NormaliserOfflineMinMax(0.0, 1.0)                                         # offline ones are rebuilt by calc_norms: fine
    assert ck.read_checkpoint(str(tmp_path), 10, off, {"node_type": off}, {"velocity": off}) is not None


def test_checkpoint_torn_between_two_saves_is_refused(tmp_path):
    """Every file is renamed into place on its own: a run killed between two renames leaves the normalisers / optimiser state of save
    k + 1 beside the parameters of save k.  The manifest (written last) does not match then, and the directory is refused."""
    import shutil
    rng = np.random.default_rng(2)
    e, n, o = _norms(rng)
    opt = ck.Adam(1e-3)
    ps = rng.normal(size=100).astype(np.float32)
    st = opt.setup(ps)
    a, b = tmp_path / "a", tmp_path / "b"
    ck.write_checkpoint(str(a), ps, e, n, o, st, ck.LossLog(), ck.LossLog())
    st2, ps2 = opt.update(st, ps, rng.normal(size=ps.size).astype(np.float32))
    e(rng.normal(size=(10, 3)).astype(np.float32))
    ck.write_checkpoint(str(b), ps2, e, n, o, st2, ck.LossLog(), ck.LossLog())
    assert ck.read_checkpoint(str(a), ps.size, *_norms(rng, False)) is not None
    for torn in (ck.CKPT_OPT, ck.CKPT_NORMS):                       # save k + 1 got as far as this file
        d = tmp_path / ("torn_" + torn)
        shutil.copytree(a, d)
        shutil.copy(b / torn, d / torn)
        with pytest.raises(ValueError, match="torn"):
            ck.read_checkpoint(str(d), ps.size, *_norms(rng, False))
    # the checksum is position-weighted: two files with the same bytes in another order differ
    f1, f2 = tmp_path / "f1", tmp_path / "f2"
    f1.write_bytes(bytes([1, 2, 3, 250]))
    f2.write_bytes(bytes([250, 3, 2, 1]))
    assert ck.file_checksum(str(f1)) == 1 + 4 + 9 + 1000 and ck.file_checksum(str(f1)) != ck.file_checksum(str(f2))


def test_julia_shim_saves_and_loads_all_four():
    """Textual pin of julia/MGNHip.jl (not executable here): save! writes parameters, log, normalisers, opt_state; load reads all four,
    restores the normalisers over the passed ones BEFORE the GraphNetwork is built from them, and returns the stored opt_state."""
    text = open(os.path.join(ROOT, "julia", "MGNHip.jl")).read()
    save = text[text.index("function save!("):text.index("# ---- graph ----")]
    load = text[text.index("function load("):text.index('"""\n`save!(')]
    assert save.index("atomically(CKPT_MANIFEST)") > save.index("atomically(CKPT_PARAMS)") and "file_checksum(f) == parse(UInt64, chk)" in load
    assert "mgn_hip_manifest.txt" in text and ck.CKPT_MANIFEST == "mgn_hip_manifest.txt"
    for const in ("CKPT_PARAMS", "CKPT_LOG", "CKPT_NORMS", "CKPT_OPT"):
        assert re.search(r"atomically\(%s\)" % const, save), const
        assert const in load, const
    for field in ("mgn.e_norm", "mgn.n_norm", "mgn.o_norm"):
        assert "snapshot(%s)" % field in save
    assert "Serialization.serialize(tmp, opt_state)" in save
    for n in ("e_norms", "n_norms", "o_norms"):
        assert re.search(r"%s = restore\(%s, stored\.%s\)" % (n, n, n[:-1]), load)
    assert load.index("restore(e_norms") < load.index("GraphNetwork(quantities")
    assert re.search(r"opt_state = Serialization\.deserialize\(ofile\)", load)
    assert "return mgn, opt_state, df_train, df_valid" in load and "return mgn, nothing" not in load
    assert "import Serialization" in text
    # the file names the two hosts share
    for name in ("mgn_hip_params.f32", "mgn_hip_log.csv"):
        assert name in text and name in (ck.CKPT_PARAMS, ck.CKPT_LOG)
