"""Build the in-tree native pieces.

  libmgn_hip.so   HIP kernels + C ABI (include/mgn_hip.h) for gfx950      [hipcc, cross-compiles w/o GPU]
  oracle/_build/libmgn_ref.so   fp32 C restatement of MGN-spec (test checker + CPU baseline)  [gcc]

Artifacts are git-ignored but travel to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmgn_hip.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_LIB = os.path.join(ORACLE_DIR, "_build", "libmgn_ref.so")

HIP_SOURCES = ["kernels.hip", "split.hip", "train.hip", "mgn_api.cpp", "mgn_train.cpp", "graph_host.cpp", "graph_prologue.cpp", "tfrecord.cpp", "comm.cpp", "graph_dev.hip"]
HIP_HEADERS = ["kernels.h", "graph_host.h", "frag.hpp", "tile_common.hpp", "split_common.hpp", "engine_internal.h", "train.h", "comm.h", "graph_dev.h", os.path.join(ROOT, "include", "mgn_hip.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; the HIP extension cannot be built")


# extra flags per source.  split.hip: see its header (the SLP vectoriser's v_pk_add_f32 inside the MFMA stream).
PER_FILE_FLAGS = {"split.hip": os.environ.get("MGN_SPLIT_FLAGS", "").split()}


def _compile_objects(objdir, extra, force, deps, verbose, per_file=None):
    """hipcc -c of every source into objdir (in parallel); returns the object paths."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    hipcc = hipcc_path()
    common = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include")] + list(extra)
    jobs, objs = [], []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        o = os.path.join(objdir, s + ".o")
        objs.append(o)
        if force or _newer(o, deps):
            pf = (per_file or {}).get(s, PER_FILE_FLAGS.get(s, []))
            jobs.append([hipcc] + common + list(pf) + ["-c", src, "-o", o])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    return objs


def build_variant(name, flags, verbose=False, per_file=None):
    """A/B experiments: compile the library with extra -D flags into lib/variants/<name>.so.
    per_file: {source: [flags]} replaces PER_FILE_FLAGS.  Without global flags only the sources that have per-file flags are
    compiled; the other objects are the default build's (lib/_obj, brought up to date first)."""
    vdir = os.path.join(LIBDIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    out = os.path.join(vdir, name + ".so")
    objdir = os.path.join(vdir, "_obj_" + name)
    if not flags and per_file:
        build_hip(force=False, verbose=verbose)
        os.makedirs(objdir, exist_ok=True)
        objs = []
        for s in HIP_SOURCES:
            if per_file.get(s):
                o = os.path.join(objdir, s + ".o")
                cmd = [hipcc_path(), "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include")] + list(per_file[s]) + ["-c", os.path.join(CSRC, s), "-o", o]
                if verbose:
                    print("[build]", " ".join(cmd), flush=True)
                subprocess.check_call(cmd)
                objs.append(o)
            else:
                objs.append(os.path.join(LIBDIR, "_obj", s + ".o"))
    else:
        objs = _compile_objects(objdir, flags, True, [], verbose, per_file)
    cmd = [hipcc_path(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out] + objs + ["-ldl", "-lrt", "-lpthread"]
    subprocess.check_call(cmd)
    shutil.rmtree(objdir, ignore_errors=True)
    return out


def _flag_stamp(extra):
    """What the objects of lib/_obj were compiled with: the global flags and every per-file flag set.  Part of the cache key --
    MGN_PROW_BLOCK changes the P / Q / CARRY row layout in kernels.hip and split.hip, and objects that disagree on it
    link fine and gather garbage (mgn_create also checks the constant of every translation unit)."""
    return repr((list(extra), sorted((k, list(v)) for k, v in PER_FILE_FLAGS.items())))


def build_hip(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    extra = os.environ.get("MGN_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DMGN_EXP_...)
    objdir = os.path.join(LIBDIR, "_obj")
    stamp_file = os.path.join(objdir, "flags.stamp")
    stamp = _flag_stamp(extra)
    have = open(stamp_file).read() if os.path.exists(stamp_file) else None
    if have != stamp and (have is not None or os.path.isdir(objdir)):
        force = True                                       # other flags than the cached objects were built with: rebuild all of them
    if not force and not _newer(LIB, deps):
        return LIB
    os.makedirs(objdir, exist_ok=True)
    if os.path.exists(stamp_file):
        os.remove(stamp_file)                              # (an interrupted build leaves no stamp: the next one starts over)
    objs = _compile_objects(objdir, extra, force, deps, verbose)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    cmd = [hipcc_path(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-ldl", "-lrt", "-lpthread"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_ref(force=False, verbose=True):
    src = os.path.join(ORACLE_DIR, "mgn_ref.c")
    if not os.path.exists(src):
        return None
    if not force and not _newer(REF_LIB, [src]):
        return REF_LIB
    os.makedirs(os.path.dirname(REF_LIB), exist_ok=True)
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-o", REF_LIB, src, "-lm"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return REF_LIB


def build_all(force=False, verbose=True):
    return build_hip(force, verbose), build_ref(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("ok")
