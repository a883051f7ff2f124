"""Synthetic meshes of SURVEY.md section 8(d) (no dataset exists: data/ is git-ignored in the
reference, .gitignore:1-2).  NumPy only; deterministic for a given seed.

M-cyl : cylinder_flow-like planar triangulation, ~2k nodes / ~12k directed edges   (cfg-1/2/5)
M-1M  : nx x ny jittered grid, random diagonal per quad; 1000x1000 -> N=1 000 000, E=5 992 002 (cfg-4)
M-flag: 40 x 40 cloth grid folded in world space, mesh edges + world (radius) edges          (cfg-3)
"""
from __future__ import annotations

import numpy as np


def cells_to_edges(cells: np.ndarray):
    """Two-way unique edge list of a triangulation (contract of GraphNetCore.triangles_to_edges as
    used at reference src/graph.jl:30).  Vectorised; first-occurrence order is not needed by the
    engine (it is edge-permutation invariant), so edges come out sorted by (max,min) key."""
    cells = np.asarray(cells, dtype=np.int64)
    e = np.concatenate([cells[:, 0:2], cells[:, 1:3], cells[:, [2, 0]]], 0)
    hi, lo = e.max(1), e.min(1)
    n = int(e.max()) + 1
    key = np.unique(hi * n + lo)
    a, b = (key // n).astype(np.int32), (key % n).astype(np.int32)
    return np.concatenate([a, b]), np.concatenate([b, a])


def grid_mesh(nx: int, ny: int, seed: int = 1234, lx: float = 1.0, ly: float = 1.0, jitter: float = 0.25):
    """nx x ny jittered grid, one random diagonal per quad.  Row-major node numbering (iy*nx+ix)."""
    rng = np.random.default_rng(seed)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    hx, hy = lx / (nx - 1), ly / (ny - 1)
    pos = np.stack([ix * hx, iy * hy], -1).reshape(-1, 2).astype(np.float64)
    jit = (rng.random((nx * ny, 2)) - 0.5) * 2.0 * jitter * np.array([hx, hy])
    interior = ((ix > 0) & (ix < nx - 1) & (iy > 0) & (iy < ny - 1)).reshape(-1)
    pos[interior] += jit[interior]
    qx, qy = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), indexing="xy")
    n00 = (qy * nx + qx).reshape(-1)
    n10, n01, n11 = n00 + 1, n00 + nx, n00 + nx + 1
    flip = rng.random(n00.size) < 0.5
    t1 = np.where(flip[:, None], np.stack([n00, n10, n01], 1), np.stack([n00, n10, n11], 1))
    t2 = np.where(flip[:, None], np.stack([n10, n11, n01], 1), np.stack([n00, n11, n01], 1))
    cells = np.concatenate([t1, t2], 0).astype(np.int32)
    return pos.astype(np.float32), cells


def mesh_1m(seed: int = 1234, nx: int = 1000, ny: int = 1000):
    """M-1M: returns (mesh_pos [N][2] f32, senders [E] i32, receivers [E] i32), 0-based."""
    pos, cells = grid_mesh(nx, ny, seed)
    s, r = cells_to_edges(cells)
    return pos, s, r


def mesh_cyl(seed: int = 1234, n_points: int = 2000):
    """M-cyl: Delaunay of n_points uniform points in [0,1.6]x[0,0.41] when scipy is present
    (2000 pts -> 11 954 directed edges), else the 64x30 jittered-grid fallback (N=1920).
    Returns (mesh_pos, cells, node_type [N] i32, velocity [N][2] f32)."""
    rng = np.random.default_rng(seed)
    try:
        from scipy.spatial import Delaunay
        pos = rng.random((n_points, 2)) * np.array([1.6, 0.41])
        cells = Delaunay(pos).simplices.astype(np.int32)
        pos = pos.astype(np.float32)
    except Exception:  # pragma: no cover - scipy is in the image
        pos, cells = grid_mesh(64, 30, seed, 1.6, 0.41)
    x, y = pos[:, 0], pos[:, 1]
    node_type = np.zeros(pos.shape[0], np.int32)
    node_type[(y < 0.012) | (y > 0.41 - 0.012) | (np.hypot(x - 0.33, y - 0.2) < 0.05)] = 6
    node_type[x < 0.02] = 4
    node_type[x > 1.6 - 0.02] = 5
    sprinkle = rng.random(pos.shape[0]) < 0.01
    node_type[sprinkle & (node_type == 0)] = 1  # exercises inflow_mask's literal `1` (src/MeshGraphNets.jl:593)
    prof = 4.0 * 1.5 * y * (0.41 - y) / 0.41 ** 2
    vel = np.stack([prof * (1.0 + 0.1 * rng.standard_normal(pos.shape[0])),
                    0.05 * rng.standard_normal(pos.shape[0])], 1).astype(np.float32)
    return pos, cells, node_type, vel


def world_edges(world_pos: np.ndarray, radius: float, mesh_senders: np.ndarray, mesh_receivers: np.ndarray):
    """Radius graph in world space without self loops and without pairs already joined by a mesh edge (DeepMind
    flag/cloth world edges).  Dense O(N^2) distances up to 8 k nodes (cloth-sized meshes; the order of the pairs is part of the
    golden fixtures), a k-d tree above (pairs sorted by (sender, receiver): the same order)."""
    wp = np.asarray(world_pos, np.float64)
    n = wp.shape[0]
    if n > 8192:
        from scipy.spatial import cKDTree
        pairs = cKDTree(wp).query_pairs(radius, output_type="ndarray")          # i < j, distance <= radius
        d2p = ((wp[pairs[:, 0]] - wp[pairs[:, 1]]) ** 2).sum(-1)
        pairs = pairs[d2p < radius * radius]
        both = np.concatenate([pairs, pairs[:, ::-1]], 0).astype(np.int64)
        key = both[:, 0] * n + both[:, 1]
        mesh_key = np.asarray(mesh_senders, np.int64) * n + np.asarray(mesh_receivers, np.int64)
        key = np.setdiff1d(key, mesh_key)                                        # sorted: (sender, receiver) order
        return (key // n).astype(np.int32), (key % n).astype(np.int32)
    d2 = ((wp[:, None, :] - wp[None, :, :]) ** 2).sum(-1)
    close = d2 < radius * radius
    np.fill_diagonal(close, False)
    close[np.asarray(mesh_senders), np.asarray(mesh_receivers)] = False
    s, r = np.nonzero(close)
    return s.astype(np.int32), r.astype(np.int32)


def mesh_flag(seed: int = 1234, nx: int = 40, ny: int = 40, radius: float = 0.045):
    """M-flag (BASELINE cfg-3): nx x ny cloth grid, folded in half in world space so that the two
    layers (0.03 apart) see each other through world edges.  Returns a dict with mesh_pos [N][2], world_pos [N][3],
    mesh edges (s, r), world edges (s2, r2), edge features ef [E][7] = (rel world 3, norm, rel mesh 2, norm),
    ef2 [E2][4] = (rel world 3, norm), node_type [N] (0 normal, 3 handle on the x = 0 column) and velocity [N][3]."""
    rng = np.random.default_rng(seed)
    pos, cells = grid_mesh(nx, ny, seed)
    s, r = cells_to_edges(cells)
    u, v = pos[:, 0].astype(np.float64), pos[:, 1].astype(np.float64)
    fold = np.abs(u - 0.5)
    world = np.stack([fold, v, 0.015 * np.sign(u - 0.5) + 0.01 * np.sin(6.0 * v) * fold], 1)
    world += 0.002 * rng.standard_normal(world.shape)
    s2, r2 = world_edges(world, radius, s, r)
    relw = world[s] - world[r]
    relm = pos[s].astype(np.float64) - pos[r].astype(np.float64)
    ef = np.concatenate([relw, np.linalg.norm(relw, axis=1, keepdims=True), relm, np.linalg.norm(relm, axis=1, keepdims=True)], 1)
    relw2 = world[s2] - world[r2]
    ef2 = np.concatenate([relw2, np.linalg.norm(relw2, axis=1, keepdims=True)], 1)
    node_type = np.zeros(pos.shape[0], np.int32)
    node_type[pos[:, 0] < 1e-6] = 3
    vel = (0.1 * rng.standard_normal((pos.shape[0], 3))).astype(np.float32)
    return dict(mesh_pos=pos, world_pos=world.astype(np.float32), cells=cells, s=s, r=r, s2=s2, r2=r2,
                ef=ef.astype(np.float32), ef2=ef2.astype(np.float32), node_type=node_type, velocity=vel)


def random_graph(n: int, e: int, seed: int = 0, allow_isolated: bool = True):
    """Small ragged test graph: random directed edges (duplicates and self loops allowed)."""
    rng = np.random.default_rng(seed)
    s = rng.integers(0, n, size=e).astype(np.int32)
    hi = max(1, n - (n // 8 if allow_isolated else 0))
    r = rng.integers(0, hi, size=e).astype(np.int32)
    return s, r
