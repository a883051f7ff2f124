"""Shared helpers for the parity tests (test infrastructure; may import the oracle)."""
import numpy as np

import mgn_amd
from mgn_amd import synth
import mgn_oracle as orc

# Stated tolerances: fp32 engine vs float64 oracle, max|d| / max|ref|
TOL_STEP = 2e-5       # latents after one processor step
TOL_15 = 1e-4         # latents / output after 15 steps
TOL_ROLLOUT = 1e-3    # relative L2 after a 100-step Euler rollout


def rel_max(a, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


def cfg_dict(Fn=9, Fe=3, O=2, L=128, mps=15):
    return dict(Fn=Fn, Fe=Fe, O=O, L=L, hidden_layers=2, mps=mps)


def make_params(cfg, seed=1234, jitter=0.1):
    return orc.init_params(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], cfg["hidden_layers"], cfg["mps"], seed, jitter)


def small_mesh(nx=8, ny=6, seed=3):
    pos, cells = synth.grid_mesh(nx, ny, seed)
    s, r = synth.cells_to_edges(cells)
    return pos, s, r


def random_inputs(N, E, cfg, seed=0):
    rng = np.random.default_rng(seed)
    nf = rng.standard_normal((N, cfg["Fn"])).astype(np.float32)
    ef = rng.standard_normal((E, cfg["Fe"])).astype(np.float32)
    return nf, ef


def engine_for(cfg, **kw):
    return mgn_amd.Engine(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], cfg["hidden_layers"], cfg["mps"], **kw)


def set_kernel_path(path):
    """Force a kernel family (tests only): 0 auto, 1 LDS-resident persistent, 2 all-streaming, 3 cooperative."""
    lib = mgn_amd.load()
    lib.mgn_debug_kernel_path.restype = __import__("ctypes").c_int
    lib.mgn_debug_kernel_path.argtypes = [__import__("ctypes").c_int]
    return lib.mgn_debug_kernel_path(path)


def set_c16_row_tiles(rt):
    """16-edge tiles per block of the small-graph edge kernel (tests only): 0 chosen by size, 1..3 pinned.  Returns the old value."""
    lib = mgn_amd.load()
    lib.mgn_debug_c16_row_tiles.restype = __import__("ctypes").c_int
    lib.mgn_debug_c16_row_tiles.argtypes = [__import__("ctypes").c_int]
    return lib.mgn_debug_c16_row_tiles(rt)


def set_fp32_split(on):
    """Large fp32 launches: 1 = the split path (bf16 matrix cores at fp32 accuracy: the default), 2 / 3 = its other edge kernels,
    0 = the fp32-MFMA kernels (tests only).  Returns the old value."""
    lib = mgn_amd.load()
    lib.mgn_debug_fp32_split.restype = __import__("ctypes").c_int
    lib.mgn_debug_fp32_split.argtypes = [__import__("ctypes").c_int]
    return lib.mgn_debug_fp32_split(on)


def set_c16_split(on):
    """16-row cooperative kernels (small meshes) on the split path, bit mask: 1 = the edge kernel at two or three row tiles per block,
    2 = the node kernel (default 3 = both), 4 = the edge kernel at one row tile too; 0 = fp32-MFMA arithmetic.  Returns the old value."""
    lib = mgn_amd.load()
    lib.mgn_debug_c16_split.restype = __import__("ctypes").c_int
    lib.mgn_debug_c16_split.argtypes = [__import__("ctypes").c_int]
    return lib.mgn_debug_c16_split(on)


def set_split_f16(on):
    """Split path: 1 (default) = two fp16 pieces per operand, three piece products (k_edge_ring_h), 0 = three bf16 pieces, six products
    (k_edge_ring).  Returns the old value."""
    lib = mgn_amd.load()
    lib.mgn_debug_split_f16.restype = __import__("ctypes").c_int
    lib.mgn_debug_split_f16.argtypes = [__import__("ctypes").c_int]
    return lib.mgn_debug_split_f16(on)


def set_renumber(mode):
    """Node numbering policy of the next set_graph calls: 0 never, 1 auto (default), 2 always breadth-first (tests only)."""
    lib = mgn_amd.load()
    lib.mgn_debug_renumber.restype = __import__("ctypes").c_int
    lib.mgn_debug_renumber.argtypes = [__import__("ctypes").c_int]
    return lib.mgn_debug_renumber(mode)


def renumbered(eng):
    eng.lib.mgn_debug_renumbered.restype = __import__("ctypes").c_int
    eng.lib.mgn_debug_renumbered.argtypes = [__import__("ctypes").c_void_p]
    return bool(eng.lib.mgn_debug_renumbered(eng.h))


def scatter_labels(pos, s, r, seed=0):
    """The same mesh under arbitrary node labels (what DeepMind's trajectories carry; create_base_graph passes them through,
    reference src/graph.jl:30-36).  Returns (pos', s', r', perm) with perm[old] = new label; per-node arrays go x' = x[argsort(perm)]."""
    N = pos.shape[0]
    perm = np.random.default_rng(seed).permutation(N).astype(np.int32)
    return pos[np.argsort(perm)], perm[s], perm[r], perm
