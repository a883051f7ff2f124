"""The Julia boundary, checked without Julia: every `ccall` of julia/*.jl is parsed and compared with the prototype of the same symbol
in include/mgn_hip.h (symbol exists, argument count, C type of every argument and of the result, as many values passed as types
declared), and the two struct mirrors (MgnConfig, MgnRolloutDesc) are compared with the C structs field by field, in order.  The shim
cannot be executed here (no julia in the image): this pins the part of it that a typo breaks silently -- a ccall with a wrong
signature does not fail, it corrupts memory."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mgn_hip.h")
JULIA_DIR = os.path.join(ROOT, "julia")


# ---- the C side -------------------------------------------------------------------------------------------------------------------
def _strip_c_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)       # preprocessor lines


_C_SCALAR = {"int": "i32", "int32_t": "i32", "int64_t": "i64", "size_t": "size", "uint64_t": "u64", "uint32_t": "u32", "float": "f32",
             "double": "f64", "uint8_t": "u8", "char": "char", "void": "void"}


def _c_type(decl):
    """'const float* nf' / 'double ms_avg[8]' / 'mgn_handle** out' -> (base class, pointer depth)"""
    d = decl.strip()
    depth = d.count("*") + (1 if "[" in d else 0)
    d = re.sub(r"\[[^\]]*\]", " ", d.replace("*", " "))
    words = [w for w in d.split() if w not in ("const", "struct")]
    base = words[0]
    return _C_SCALAR.get(base, base), depth


def c_prototypes():
    text = _strip_c_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mgn_\w+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "return" in ret:
            continue
        arglist = [] if args in ("", "void") else [_c_type(a) for a in args.split(",")]
        protos[name] = (_c_type(ret + " x"), arglist)
    return protos


def c_struct(name):
    text = _strip_c_comments(open(HEADER).read())
    m = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), text, flags=re.S)
    assert m, name
    fields = []
    for stmt in m.group(1).split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        first, *rest = [p.strip() for p in stmt.split(",")]
        base, depth = _c_type(first)
        fname = re.sub(r"\[[^\]]*\]", "", first.replace("*", " ")).split()[-1]
        fields.append((fname, (base, depth)))
        for r in rest:                                    # `int32_t a, b, c;`: the later declarators share the base type
            fields.append((r.replace("*", "").strip(), (base, r.count("*"))))
    return fields


# ---- the Julia side ---------------------------------------------------------------------------------------------------------------
_JL = {"Cint": ("i32", 0), "Int32": ("i32", 0), "Int64": ("i64", 0), "Csize_t": ("size", 0), "UInt64": ("u64", 0), "UInt32": ("u32", 0),
       "Float32": ("f32", 0), "Float64": ("f64", 0), "Cvoid": ("void", 0), "Cstring": ("char", 1),
       "Ptr{Cvoid}": ("void", 1), "Ptr{Float32}": ("f32", 1), "Ref{Float32}": ("f32", 1), "Ptr{Int32}": ("i32", 1), "Ptr{Int64}": ("i64", 1),
       "Ptr{Float64}": ("f64", 1), "Ptr{UInt8}": ("u8", 1), "Ref{MgnConfig}": ("mgn_config", 1), "Ref{MgnRolloutDesc}": ("mgn_rollout_desc", 1),
       "Ref{Ptr{Cvoid}}": ("void", 2)}


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _strip_jl_comments(text):
    text = re.sub(r"#=.*?=#", " ", text, flags=re.S)
    lines = []
    for line in text.splitlines():
        # a '#' outside a string starts a comment (the shim has no '#' inside strings except in docstrings, which hold no ccall)
        lines.append(line.split("#")[0] if '"' not in line.split("#")[0] or line.split("#")[0].count('"') % 2 == 0 else line)
    return "\n".join(lines)


def julia_ccalls(path):
    text = _strip_jl_comments(open(path).read())
    calls = []
    for m in re.finditer(r"ccall\(", text):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        parts = _split_top(text[m.end():i - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*LIB\s*\)", parts[0])
        assert sym, parts[0]
        types = _split_top(parts[2].strip()[1:-1])
        calls.append((sym.group(1), parts[1], [t for t in types if t], parts[3:]))
    return calls


def julia_struct(path, name):
    text = _strip_jl_comments(open(path).read())
    m = re.search(r"(?:mutable\s+)?struct\s+%s\b(.*?)\nend" % name, text, flags=re.S)
    assert m, name
    return [(f, t) for f, t in re.findall(r"(\w+)::([\w\{\}]+)", m.group(1))]


def _compatible(jl, c):
    if jl == c:
        return True
    (jb, jd), (cb, cd) = jl, c
    if jd != cd:
        return False
    if jd >= 1 and jb == "void":                 # Ptr{Cvoid}: an opaque handle or a void* -- not a typed data pointer
        return cb in ("void", "mgn_handle", "mgn_tfrecord", "mgn_engine")
    if jd == 1 and jb == "u8" and cb == "void":  # raw bytes (the communicator id)
        return True
    return False


# ---- tests ------------------------------------------------------------------------------------------------------------------------
def test_every_ccall_matches_the_header():
    protos = c_prototypes()
    assert len(protos) >= 70 and "mgn_forward" in protos and protos["mgn_forward"][1] == [("mgn_handle", 1), ("f32", 1), ("f32", 1), ("f32", 1)]
    files = sorted(f for f in os.listdir(JULIA_DIR) if f.endswith(".jl"))
    seen = set()
    ncalls = 0
    for f in files:
        for sym, ret, types, args in julia_ccalls(os.path.join(JULIA_DIR, f)):
            where = f"{f}: ccall :{sym}"
            assert sym in protos, f"{where}: not declared in include/mgn_hip.h"
            cret, cargs = protos[sym]
            assert ret in _JL, f"{where}: unknown result type {ret}"
            assert _compatible(_JL[ret], cret), f"{where}: result {ret} vs C {cret}"
            assert len(types) == len(cargs), f"{where}: {len(types)} argument types, the header declares {len(cargs)}"
            assert len(args) == len(types), f"{where}: {len(args)} values passed for {len(types)} argument types"
            for k, (t, c) in enumerate(zip(types, cargs)):
                assert t in _JL, f"{where}: unknown argument type {t}"
                assert _compatible(_JL[t], c), f"{where}: argument {k + 1} is {t}, the header says {c}"
            seen.add(sym)
            ncalls += 1
    # the front door: what src/graph.jl / src/solve.jl / src/strategies.jl reach through the shim
    for need in ("mgn_create", "mgn_destroy", "mgn_set_params", "mgn_set_graph", "mgn_forward", "mgn_step", "mgn_set_norms", "mgn_rollout",
                 "mgn_ode_vjp", "mgn_forward_vjp", "mgn_ode_step", "mgn_set_static", "mgn_param_count", "mgn_last_error"):
        assert need in seen, f"the shim does not bind {need}"
    assert ncalls >= 20


def test_abi_version_constants_agree():
    hv = int(re.search(r"#define\s+MGN_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    jv = int(re.search(r"const ABI_VERSION = (\d+)", open(os.path.join(JULIA_DIR, "MGNHip.jl")).read()).group(1))
    pv = int(re.search(r"^ABI_VERSION = (\d+)", open(os.path.join(ROOT, "meshgraphnets.jl_amd", "_capi.py")).read(), flags=re.M).group(1))
    assert hv == jv == pv


def test_struct_mirrors_match_field_by_field():
    path = os.path.join(JULIA_DIR, "MGNHip.jl")
    for jl_name, c_name in (("MgnConfig", "mgn_config"), ("MgnRolloutDesc", "mgn_rollout_desc")):
        cf = c_struct(c_name)
        jf = julia_struct(path, jl_name)
        assert [n for n, _ in jf] == [n for n, _ in cf], (jl_name, [n for n, _ in jf], [n for n, _ in cf])
        for (n, jt), (_, ct) in zip(jf, cf):
            assert jt in _JL, (jl_name, n, jt)
            assert _JL[jt] == ct or (ct[1] == 1 and _JL[jt] == ct), (jl_name, n, jt, ct)


def test_python_mirrors_match_the_header_too():
    """The ctypes twins (the tested host) against the same C structs: a field added to the header must be added to both mirrors."""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    import mgn_amd
    capi = mgn_amd._capi if hasattr(mgn_amd, "_capi") else __import__("importlib").import_module("mgn_amd._capi")
    ct = {C.c_int32: ("i32", 0), C.c_float: ("f32", 0), C.c_double: ("f64", 0), C.POINTER(C.c_float): ("f32", 1), C.POINTER(C.c_uint8): ("u8", 1)}
    for cls, c_name in ((capi.MgnConfig, "mgn_config"), (capi.MgnRolloutDesc, "mgn_rollout_desc")):
        cf = c_struct(c_name)
        assert [n for n, _ in cls._fields_] == [n for n, _ in cf], c_name
        for (n, t), (_, want) in zip(cls._fields_, cf):
            assert ct[t] == want, (c_name, n, t, want)


def test_step_returns_a_tuple_and_names_do_not_clash():
    """Two textual properties INTEGRATION.md promises: step! returns ((gs,), loss) so that the update loop at
    src/MeshGraphNets.jl:375-377 runs unchanged, and GraphNetCore is imported selectively (no `using GraphNetCore` beside the shim's
    own GraphNetwork / FeatureGraph / step! / load / save!)."""
    text = _strip_jl_comments(open(os.path.join(JULIA_DIR, "MGNHip.jl")).read())
    assert re.search(r"return \(gs,\), loss\[\]", text)
    assert not re.search(r"^\s*using\s+GraphNetCore", text, flags=re.M)
    imp = re.search(r"import GraphNetCore:(.*?)\n\S", text, flags=re.S).group(1)
    for name in ("one_hot", "triangles_to_edges", "parse_edges", "mse_reduce", "inverse_data", "NormaliserOnline"):
        assert name in imp
    for name in ("GraphNetwork", "FeatureGraph", "step!", "load", "save!"):
        assert name not in imp and re.search(r"export[^\n]*\b%s" % re.escape(name), text)
    assert "ChainRulesCore.rrule(::typeof(forward)" in text
