"""ctypes loader for oracle/mgn_ref.c (fp32 C restatement).  TEST INFRASTRUCTURE / CPU BASELINE ONLY."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmgn_ref.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} missing: run __graft_entry__.build()")
        _lib = C.CDLL(LIB)
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        _lib.mgn_ref_processor_steps.argtypes = [f32p] + [C.c_int] * 6 + [C.c_int64, i32p, i32p, f32p, f32p, C.c_int]
        _lib.mgn_ref_forward.argtypes = [f32p] + [C.c_int] * 6 + [C.c_int64, i32p, i32p, f32p, f32p, f32p]
        _lib.mgn_ref_num_threads.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def num_threads():
    return int(load().mgn_ref_num_threads())


def processor_steps(params, cfg, v, e, senders, receivers, nsteps):
    lib = load()
    params = np.ascontiguousarray(params, np.float32)
    v = np.array(v, np.float32, order="C")
    e = np.array(e, np.float32, order="C")
    s = np.ascontiguousarray(senders, np.int32)
    r = np.ascontiguousarray(receivers, np.int32)
    rc = lib.mgn_ref_processor_steps(_p(params, C.c_float), cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], cfg["mps"],
                                     v.shape[0], e.shape[0], _p(s, C.c_int32), _p(r, C.c_int32),
                                     _p(v, C.c_float), _p(e, C.c_float), nsteps)
    if rc != 0:
        raise RuntimeError("mgn_ref_processor_steps failed")
    return v, e


def forward(params, cfg, nf, ef, senders, receivers):
    lib = load()
    params = np.ascontiguousarray(params, np.float32)
    nf = np.ascontiguousarray(nf, np.float32)
    ef = np.ascontiguousarray(ef, np.float32)
    s = np.ascontiguousarray(senders, np.int32)
    r = np.ascontiguousarray(receivers, np.int32)
    out = np.zeros((nf.shape[0], cfg["O"]), np.float32)
    rc = lib.mgn_ref_forward(_p(params, C.c_float), cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], cfg["mps"],
                             nf.shape[0], ef.shape[0], _p(s, C.c_int32), _p(r, C.c_int32),
                             _p(nf, C.c_float), _p(ef, C.c_float), _p(out, C.c_float))
    if rc != 0:
        raise RuntimeError("mgn_ref_forward failed")
    return out
