"""Checkpoint state of the GraphNetCore-shaped surface: what `load` / `save!` carry between runs.

The reference's call sites (src/MeshGraphNets.jl:282-285, 324-325, 460-471, 537-540) assume that a checkpoint brings back FOUR things:
the parameters, the loss log (`step = last(df_train.step)` on resume), the optimiser state (`Optimisers.setup` only when `load`
returned `nothing`, :287-289) and the NORMALISERS -- `eval_network` builds fresh ones with `calc_norms` and relies on `load` to put the
trained statistics back (:529-540); a cylinder_flow model normalises velocity and edges with `NormaliserOnline` (:92,193-199), whose
statistics exist nowhere else.  This module is the tested Python twin of that part of julia/MGNHip.jl (`snapshot`, `restore`, the four
files); `engine.load` / `engine.save` are the call-shaped wrappers.

Files in a checkpoint directory: `mgn_hip_params.f32` (packed parameters, raw little-endian float32) and `mgn_hip_log.csv`
(kind,step,loss) are the SAME files the Julia shim writes; the normalisers and the optimiser state are `.npz` here and Julia
`Serialization` files there (neither side can read the other's object format).  `mgn_hip_manifest.txt`, written last, lists
size and checksum of the other four: each file is renamed into place on its own, so a run killed between two renames leaves files of
two saves side by side -- the manifest then does not match and `read_checkpoint` refuses the directory instead of resuming with the
Adam moments of step k + 1 over the parameters of step k.
"""
from __future__ import annotations

import os

import numpy as np

CKPT_PARAMS = "mgn_hip_params.f32"
CKPT_LOG = "mgn_hip_log.csv"
CKPT_NORMS = "mgn_hip_norms.npz"
CKPT_OPT = "mgn_hip_opt_state.npz"
CKPT_MANIFEST = "mgn_hip_manifest.txt"

_SKIP_FIELDS = ("engine",)          # a NormaliserOnline may hold the Engine it reduces on: a handle, not state


class LossLog:
    """The two columns the reference reads of `df_train` / `df_valid` (src/MeshGraphNets.jl:324-330,383)."""

    def __init__(self):
        self.step, self.loss = [], []


class Adam:
    """Optimisers.Adam(eta, (beta1, beta2), epsilon) on the packed vector: `setup(ps)` -> state, `update(state, ps, g)` ->
    (state, ps') as at src/MeshGraphNets.jl:288,376 -- enough of an optimiser for the twin's round-trip tests."""

    def __init__(self, eta=1e-3, beta=(0.9, 0.999), epsilon=1e-8):
        self.eta, self.beta, self.epsilon = float(eta), (float(beta[0]), float(beta[1])), float(epsilon)

    def setup(self, ps):
        return {"mt": np.zeros_like(ps, dtype=np.float32), "vt": np.zeros_like(ps, dtype=np.float32),
                "beta_t": np.array(self.beta, np.float64)}

    def update(self, state, ps, g):
        b1, b2 = self.beta
        g = np.asarray(g, np.float32)
        mt = (b1 * state["mt"] + (1 - b1) * g).astype(np.float32)
        vt = (b2 * state["vt"] + (1 - b2) * g * g).astype(np.float32)
        bt = state["beta_t"]
        dx = mt / (1 - bt[0]) / (np.sqrt(vt / (1 - bt[1])) + self.epsilon) * self.eta
        return {"mt": mt, "vt": vt, "beta_t": bt * np.array(self.beta)}, (ps - dx).astype(np.float32)


# ---- normalisers <-> plain data ----------------------------------------------------------------------------------------------------
def snapshot(obj, prefix, out):
    """Flatten a normaliser (or a dict of them) into `out[prefix...] = array`; class names under `<prefix>/__type__`."""
    if isinstance(obj, dict):
        out[prefix + "/__keys__"] = np.array(sorted(obj), dtype=str)
        for k in obj:
            snapshot(obj[k], prefix + "/" + k, out)
        return out
    out[prefix + "/__type__"] = np.array(type(obj).__name__)
    for f, v in vars(obj).items():
        if f in _SKIP_FIELDS or callable(v):
            continue
        out[prefix + "/" + f] = np.asarray(v)
    return out


def restore(template, prefix, data):
    """Put the stored fields over a freshly built normaliser of the same kind (what `calc_norms` hands to `load`); a kind mismatch is
    an error, not a silent pick.  Dicts are restored key by key; a stored key the template lacks is an error too."""
    if isinstance(template, dict):
        keys = [str(k) for k in data[prefix + "/__keys__"]]
        missing = [k for k in keys if k not in template]
        if missing:
            raise ValueError("checkpoint holds normalisers for %s, load was handed %s" % (keys, sorted(template)))
        for k in keys:
            template[k] = restore(template[k], prefix + "/" + k, data)
        return template
    kind = str(data[prefix + "/__type__"])
    if type(template).__name__ != kind:
        raise ValueError("checkpoint holds a %s at %s, load was handed a %s: build the normalisers as the training run did"
                         % (kind, prefix, type(template).__name__))
    for f, cur in vars(template).items():
        key = prefix + "/" + f
        if f in _SKIP_FIELDS or callable(cur) or key not in data:
            continue
        v = data[key]
        if isinstance(cur, np.ndarray):
            setattr(template, f, v.astype(cur.dtype).reshape(v.shape))
        else:
            setattr(template, f, type(cur)(v[()]) if v.ndim == 0 else v)
    return template


def _has_online(n):
    if isinstance(n, dict):
        return any(_has_online(v) for v in n.values())
    return type(n).__name__ == "NormaliserOnline"


def _atomic(path, name, write):
    tmp = os.path.join(path, name + ".tmp")
    write(tmp)
    os.replace(tmp, os.path.join(path, name))


def file_checksum(fname):
    """sum over the file's bytes b_i (i from 1) of i * b_i, modulo 2^64: the same three lines in julia/MGNHip.jl."""
    b = np.fromfile(fname, np.uint8).astype(np.uint64)
    with np.errstate(over="ignore"):
        return int((np.arange(1, b.size + 1, dtype=np.uint64) * b).sum(dtype=np.uint64))


def _manifest_lines(path, names):
    return ["%s,%d,%d\n" % (n, os.path.getsize(os.path.join(path, n)), file_checksum(os.path.join(path, n))) for n in names]


# ---- the four files ----------------------------------------------------------------------------------------------------------------
def write_checkpoint(path, ps, e_norm, n_norm, o_norm, opt_state, df_train, df_valid):
    """Everything `save!` is given.  Each file is written beside its target and renamed, the manifest (sizes and checksums of the four)
    last: a run killed inside leaves either the previous checkpoint whole or a directory whose manifest does not match, which
    `read_checkpoint` refuses."""
    os.makedirs(path, exist_ok=True)
    norms = {}
    snapshot(e_norm, "e_norm", norms)
    snapshot(n_norm, "n_norm", norms)
    snapshot(o_norm, "o_norm", norms)

    def w_norms(tmp):
        with open(tmp, "wb") as f:
            np.savez(f, **norms)

    def w_opt(tmp):
        with open(tmp, "wb") as f:
            np.savez(f, __none__=np.array(opt_state is None), **({} if opt_state is None else opt_state))

    def w_log(tmp):
        with open(tmp, "w") as f:
            for kind, log in (("train", df_train), ("valid", df_valid)):
                for s, l in zip(log.step, log.loss):
                    f.write("%s,%d,%r\n" % (kind, int(s), float(np.float32(l))))

    _atomic(path, CKPT_NORMS, w_norms)
    _atomic(path, CKPT_OPT, w_opt)
    _atomic(path, CKPT_LOG, w_log)
    _atomic(path, CKPT_PARAMS, lambda tmp: np.ascontiguousarray(ps, "<f4").tofile(tmp))

    def w_manifest(tmp):
        with open(tmp, "w") as f:
            f.writelines(_manifest_lines(path, (CKPT_NORMS, CKPT_OPT, CKPT_LOG, CKPT_PARAMS)))

    _atomic(path, CKPT_MANIFEST, w_manifest)


def read_checkpoint(path, nparams, e_norm, n_norm, o_norm, want_opt_state=True):
    """-> None when `path` holds no checkpoint, else (ps, e_norm, n_norm, o_norm, opt_state, df_train, df_valid) with the stored
    statistics restored over the passed normalisers.  Parameters without a normaliser file (written before those were stored) are
    refused when an online normaliser was passed: it would evaluate with empty statistics."""
    pfile = os.path.join(path, CKPT_PARAMS)
    if not os.path.isfile(pfile):
        return None
    if os.path.getsize(pfile) != 4 * nparams:
        raise ValueError("checkpoint %s holds %d bytes, this model has %d" % (pfile, os.path.getsize(pfile), 4 * nparams))
    mfile = os.path.join(path, CKPT_MANIFEST)
    if os.path.isfile(mfile):          # (absent: a checkpoint written before the manifest existed -- taken as it is)
        for line in open(mfile):
            name, size, chk = line.strip().split(",")
            f = os.path.join(path, name)
            if not os.path.isfile(f) or os.path.getsize(f) != int(size) or file_checksum(f) != int(chk):
                raise ValueError("checkpoint in %s is torn: %s is not the file its manifest lists (a run was killed inside save!; the "
                                 "directory mixes two saves)" % (path, name))
    ps = np.fromfile(pfile, "<f4").astype(np.float32)
    nfile = os.path.join(path, CKPT_NORMS)
    if os.path.isfile(nfile):
        with np.load(nfile) as data:
            e_norm = restore(e_norm, "e_norm", data)
            n_norm = restore(n_norm, "n_norm", data)
            o_norm = restore(o_norm, "o_norm", data)
    elif _has_online(e_norm) or _has_online(n_norm) or _has_online(o_norm):
        raise ValueError("checkpoint in %s has no %s: its online normalisers' statistics were not saved and cannot be rebuilt"
                         % (path, CKPT_NORMS))
    opt_state = None
    ofile = os.path.join(path, CKPT_OPT)
    if want_opt_state and os.path.isfile(ofile):
        with np.load(ofile) as data:
            if not bool(data["__none__"]):
                opt_state = {k: data[k] for k in data.files if k != "__none__"}
    df_train, df_valid = LossLog(), LossLog()
    lfile = os.path.join(path, CKPT_LOG)
    if os.path.isfile(lfile):
        for line in open(lfile):
            kind, step, loss = line.strip().split(",")
            log = df_train if kind == "train" else df_valid
            log.step.append(int(step))
            log.loss.append(np.float32(loss))
    return ps, e_norm, n_norm, o_norm, opt_state, df_train, df_valid
