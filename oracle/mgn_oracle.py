"""MGN-spec v1 float64 oracle  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: the reference (una-auxme/MeshGraphNets.jl @ 2024_08_07) delegates all
arithmetic on this path to GraphNetCore.jl 0.3 / Lux 0.5, neither of which is vendored under
/root/reference (Project.toml:11,15,36,40; Manifest git-ignored .gitignore:7-8), `julia` is not
installed here, and the reference's own tests hold no numerical fixtures (test/runtests.jl:11-19
is Aqua static QA only).  This file therefore restates the *published* algorithm (DeepMind
MeshGraphNets, arXiv 2010.03409, which README.md:9-19 names as what GraphNetCore implements)
under the constraints of every reference call site, each cited below.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

All arrays are C row-major [count][feat]  ==  the bytes of Julia's column-major (feat x count).
Indices are 0-based here; the 1-based Julia boundary (src/graph.jl:31-34) is handled by callers.
"""
from __future__ import annotations

import numpy as np

LN_EPS = 1e-5  # Lux 0.5 LayerNorm default epsilon [GNC-unverified], SURVEY.md section 8c


# --------------------------------------------------------------------------------------------
# Graph prologue  (src/graph.jl:25-55 create_base_graph)
# --------------------------------------------------------------------------------------------
def one_hot(types, depth, offset=0):
    """one_hot(vec, depth, offset) as called at src/graph.jl:26-27:
    depth = type_max - type_min + 1, offset = 1 - type_min (1-based Julia) => column type-type_min."""
    types = np.asarray(types).reshape(-1)
    out = np.zeros((types.size, depth), dtype=np.float64)
    out[np.arange(types.size), types + offset] = 1.0
    return out


def triangles_to_edges(cells):
    """triangles_to_edges(cells) called at src/graph.jl:30.  cells: [C][3] int.
    Undirected unique edges of the triangulation in first-occurrence order, returned two-way:
    senders=[a;b], receivers=[b;a] with a=max(i,j), b=min(i,j)  (DeepMind common.triangles_to_edges)."""
    cells = np.asarray(cells, dtype=np.int64)
    e = np.concatenate([cells[:, 0:2], cells[:, 1:3], np.stack([cells[:, 2], cells[:, 0]], 1)], 0)
    hi = e.max(1)
    lo = e.min(1)
    key = hi * (int(e.max()) + 1) + lo
    _, first = np.unique(key, return_index=True)
    first.sort()
    a, b = hi[first], lo[first]
    return np.concatenate([a, b]), np.concatenate([b, a])


def edge_features(mesh_pos, senders, receivers):
    """edge_features = [pos[s]-pos[r]; ||pos[s]-pos[r]||]  (src/graph.jl:35-36,49-52)."""
    rel = np.asarray(mesh_pos, np.float64)[senders] - np.asarray(mesh_pos, np.float64)[receivers]
    return np.concatenate([rel, np.linalg.norm(rel, axis=1, keepdims=True)], 1)


# --------------------------------------------------------------------------------------------
# Normalisers  (constructed at src/MeshGraphNets.jl:79-203, applied at src/graph.jl:80-93,
#               inverted at src/solve.jl:205-210)
# --------------------------------------------------------------------------------------------
class NormMinMax:
    """NormaliserOfflineMinMax(dmin, dmax[, tmin, tmax]); defaults target (0,1)."""

    def __init__(self, dmin, dmax, tmin=0.0, tmax=1.0):
        self.dmin, self.dmax, self.tmin, self.tmax = map(float, (dmin, dmax, tmin, tmax))

    def __call__(self, x):
        return (x - self.dmin) / (self.dmax - self.dmin) * (self.tmax - self.tmin) + self.tmin

    def inverse(self, y):
        return (y - self.tmin) / (self.tmax - self.tmin) * (self.dmax - self.dmin) + self.dmin

    def affine(self, dim):
        """(scale, shift) per feature with forward y = x*scale + shift."""
        s = (self.tmax - self.tmin) / (self.dmax - self.dmin)
        return np.full(dim, s), np.full(dim, self.tmin - self.dmin * s)


class NormMeanStd:
    """NormaliserOfflineMeanStd(mean, std); also the frozen state of a NormaliserOnline
    (mean, max(std, 1e-8)) once accumulation has stopped (max_acc reached / inference)."""

    def __init__(self, mean, std):
        self.mean = np.asarray(mean, np.float64)
        self.std = np.maximum(np.asarray(std, np.float64), 1e-8)

    def __call__(self, x):
        return (x - self.mean) / self.std

    def inverse(self, y):
        return y * self.std + self.mean

    def affine(self, dim):
        s = np.broadcast_to(1.0 / self.std, (dim,)).copy()
        return s, -np.broadcast_to(self.mean, (dim,)) * s


class NormOnline:
    """NormaliserOnline(dims; max_acc): running sum / sum-of-squares / count over rows while
    count < max_acc, then (x-mean)/max(std,1e-8)  (src/MeshGraphNets.jl:92,193-199)."""

    def __init__(self, dim, max_acc=1e6, eps=1e-8):
        self.dim, self.max_acc, self.eps = dim, float(max_acc), eps
        self.acc_sum = np.zeros(dim)
        self.acc_sq = np.zeros(dim)
        self.count = 0.0
        self.n_acc = 0.0

    def accumulate(self, x):
        if self.n_acc < self.max_acc:
            self.acc_sum += x.sum(0)
            self.acc_sq += (x * x).sum(0)
            self.count += x.shape[0]
            self.n_acc += 1.0

    def frozen(self):
        c = max(self.count, 1.0)
        mean = self.acc_sum / c
        std = np.sqrt(np.maximum(self.acc_sq / c - mean * mean, 0.0))
        return NormMeanStd(mean, np.maximum(std, self.eps))

    def __call__(self, x):
        self.accumulate(x)
        return self.frozen()(x)

    def inverse(self, y):
        return self.frozen().inverse(y)


# --------------------------------------------------------------------------------------------
# Parameters (packing order fixed by SURVEY.md section 8b)
# --------------------------------------------------------------------------------------------
def mlp_shapes(n_in, L, n_out, hidden_layers, ln):
    dims = [n_in] + [L] * hidden_layers + [n_out]
    shapes = []
    for i in range(len(dims) - 1):
        shapes.append(("W%d" % (i + 1), (dims[i], dims[i + 1])))  # row-major [in][out]
        shapes.append(("b%d" % (i + 1), (dims[i + 1],)))
    if ln:
        shapes.append(("ln_scale", (n_out,)))
        shapes.append(("ln_bias", (n_out,)))
    return shapes


def model_layout(Fn, Fe, O, L, hidden_layers, mps, Fe2=None):
    """[(block_name, [(tensor_name, shape)...])...] in packed order:
    encoder-node, encoder-edge (per edge set), step1-edge (per edge set), step1-node, ..., decoder.
    Fe2: input width of a second edge set (MGN-spec "per edge set": world edges of flag_simple; the reference's
    FeatureGraph has one set, src/graph.jl:87-96).  The node MLP then takes [v; agg_1; agg_2]."""
    K = 2 if Fe2 else 1
    blocks = [("enc_node", mlp_shapes(Fn, L, L, hidden_layers, True)),
              ("enc_edge", mlp_shapes(Fe, L, L, hidden_layers, True))]
    if K == 2:
        blocks.append(("enc_edge2", mlp_shapes(Fe2, L, L, hidden_layers, True)))
    for k in range(mps):
        blocks.append(("proc%d_edge" % k, mlp_shapes(3 * L, L, L, hidden_layers, True)))
        if K == 2:
            blocks.append(("proc%d_edge2" % k, mlp_shapes(3 * L, L, L, hidden_layers, True)))
        blocks.append(("proc%d_node" % k, mlp_shapes((1 + K) * L, L, L, hidden_layers, True)))
    blocks.append(("decoder", mlp_shapes(L, L, O, hidden_layers, False)))
    return blocks


def param_count(Fn, Fe, O, L, hidden_layers, mps, Fe2=None):
    return sum(int(np.prod(s)) for _, ts in model_layout(Fn, Fe, O, L, hidden_layers, mps, Fe2) for _, s in ts)


def unpack_params(packed, Fn, Fe, O, L, hidden_layers, mps, Fe2=None):
    packed = np.asarray(packed)
    out, off = {}, 0
    for bname, tensors in model_layout(Fn, Fe, O, L, hidden_layers, mps, Fe2):
        d = {}
        for tname, shape in tensors:
            n = int(np.prod(shape))
            d[tname] = packed[off:off + n].reshape(shape)
            off += n
        out[bname] = d
    assert off == packed.size, (off, packed.size)
    return out


def init_params(Fn, Fe, O, L, hidden_layers, mps, seed=1234, ln_jitter=0.0, Fe2=None):
    """Glorot-uniform W, zero b, gamma=1, beta=0 (SURVEY.md 8c 'Init').  ln_jitter/bias jitter > 0
    perturbs b, gamma, beta so that tests exercise them."""
    rng = np.random.default_rng(seed)
    chunks = []
    for _, tensors in model_layout(Fn, Fe, O, L, hidden_layers, mps, Fe2):
        for tname, shape in tensors:
            if tname.startswith("W"):
                lim = np.sqrt(6.0 / (shape[0] + shape[1]))
                chunks.append(rng.uniform(-lim, lim, size=shape).ravel())
            elif tname.startswith("b"):
                chunks.append(ln_jitter * rng.standard_normal(shape).ravel())
            elif tname == "ln_scale":
                chunks.append(1.0 + ln_jitter * rng.standard_normal(shape).ravel())
            else:
                chunks.append(ln_jitter * rng.standard_normal(shape).ravel())
    return np.concatenate(chunks).astype(np.float32)


# --------------------------------------------------------------------------------------------
# Encode - Process - Decode  (mgn.model(graph, ps, st) at src/solve.jl:200)
# --------------------------------------------------------------------------------------------
# spec_variant switches (julia/spec_probe.jl tells which one GraphNetCore 0.3 / Lux 0.5 really compute).
# Module-level on purpose: every function of the oracle (forward, reverse mode, rollout) follows them; tests set and restore them.
#   LN_MODE: 0 = (x - mean) / sqrt(var + eps)  [MGN-spec v1]     1 = (x - mean) / (sqrt(var) + eps)
#   LN_DIMS: "row" = statistics per node / edge over its L features [MGN-spec v1]
#            "all" = statistics over the WHOLE array, rows included (Lux 0.5 LayerNorm(shape) with dims = Colon());
#                    the engine's ln_dims = MGN_LN_ALL.  Forward functions and the reverse mode follow it (the pullback's two means run
#                    over the same set of entries as the statistics).
LN_MODE = 0
LN_DIMS = "row"


def layer_norm(x, gamma, beta):
    ax = -1 if LN_DIMS == "row" else None
    mu = x.mean(ax, keepdims=True)
    var = ((x - mu) ** 2).mean(ax, keepdims=True)  # biased variance
    den = np.sqrt(var + LN_EPS) if LN_MODE == 0 else np.sqrt(var) + LN_EPS
    return (x - mu) / den * gamma + beta


def mlp(x, p, hidden_layers):
    for i in range(1, hidden_layers + 2):
        x = x @ p["W%d" % i] + p["b%d" % i]
        if i <= hidden_layers:
            x = np.maximum(x, 0.0)
    if "ln_scale" in p:
        x = layer_norm(x, p["ln_scale"], p["ln_bias"])
    return x


def scatter_add(rows, index, n):
    out = np.zeros((n, rows.shape[1]), rows.dtype)
    np.add.at(out, index, rows)
    return out


def encode(P, nf, ef, h):
    return mlp(nf, P["enc_node"], h), mlp(ef, P["enc_edge"], h)


def processor_step(P, k, v, e, senders, receivers, h, set2=None):
    """One message-passing step (DeepMind GraphNetBlock order): the node update consumes e'
    BEFORE the residual is added; then v += v', e += e'.
    set2 = (e2, senders2, receivers2): second edge set with its own edge MLP; v' = MLP_v([v; agg_1; agg_2])."""
    e_new = mlp(np.concatenate([v[senders], v[receivers], e], 1), P["proc%d_edge" % k], h)
    aggs = [scatter_add(e_new, receivers, v.shape[0])]
    if set2 is not None:
        e2, s2, r2 = set2
        e2_new = mlp(np.concatenate([v[s2], v[r2], e2], 1), P["proc%d_edge2" % k], h)
        aggs.append(scatter_add(e2_new, r2, v.shape[0]))
    v_new = mlp(np.concatenate([v] + aggs, 1), P["proc%d_node" % k], h)
    if set2 is not None:
        return v + v_new, e + e_new, e2 + e2_new
    return v + v_new, e + e_new


def decode(P, v, h):
    return mlp(v, P["decoder"], h)


def _unpack(packed, cfg, dtype):
    return unpack_params(np.asarray(packed, dtype), cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], cfg["hidden_layers"], cfg["mps"],
                         cfg.get("Fe2"))


def forward(packed, cfg, nf, ef, senders, receivers, return_latents=False, dtype=np.float64, set2=None):
    """cfg: dict(Fn, Fe, O, L, hidden_layers, mps[, Fe2]).  nf [N][Fn], ef [E][Fe] (already normalised,
    i.e. the FeatureGraph of src/graph.jl:87-96).  set2 = (ef2 [E2][Fe2], senders2, receivers2).  Returns out [N][O]."""
    h = cfg["hidden_layers"]
    P = _unpack(packed, cfg, dtype)
    v, e = encode(P, np.asarray(nf, dtype), np.asarray(ef, dtype), h)
    if set2 is not None:
        e2, s2, r2 = mlp(np.asarray(set2[0], dtype), P["enc_edge2"], h), set2[1], set2[2]
    lat = [(v.copy(), e.copy())]
    for k in range(cfg["mps"]):
        if set2 is not None:
            v, e, e2 = processor_step(P, k, v, e, senders, receivers, h, (e2, s2, r2))
        else:
            v, e = processor_step(P, k, v, e, senders, receivers, h)
        if return_latents:
            lat.append((v.copy(), e.copy()) if set2 is None else (v.copy(), e.copy(), e2.copy()))
    out = decode(P, v, h)
    return (out, lat) if return_latents else out


def processor_steps(packed, cfg, v, e, senders, receivers, nsteps, dtype=np.float64, set2=None):
    """The benchmarked unit (SURVEY.md 8b mgn_processor_steps): nsteps of A6 on given latents.
    set2 = (e2 [E2][L], senders2, receivers2): returns (v, e, e2)."""
    h = cfg["hidden_layers"]
    P = _unpack(packed, cfg, dtype)
    v, e = np.asarray(v, dtype), np.asarray(e, dtype)
    if set2 is not None:
        e2, s2, r2 = np.asarray(set2[0], dtype), set2[1], set2[2]
        for k in range(nsteps):
            v, e, e2 = processor_step(P, k, v, e, senders, receivers, h, (e2, s2, r2))
        return v, e, e2
    for k in range(nsteps):
        v, e = processor_step(P, k, v, e, senders, receivers, h)
    return v, e


# --------------------------------------------------------------------------------------------
# Training step  (GraphNetCore.step!(mgn, graph, target, mask, mse_reduce) as called at
#                 src/strategies.jl:418-422 and consumed at src/MeshGraphNets.jl:370-378)
# Reverse-mode differentiation written out by hand (the reference uses Zygote); float64.
# --------------------------------------------------------------------------------------------
def mse_reduce(target, out):
    """mse_reduce(target, output): sum of squared differences over the O rows per node (src/strategies.jl:421)."""
    return ((np.asarray(target) - np.asarray(out)) ** 2).sum(1)


def _mlp_fwd(x, p, h):
    """MLP forward keeping what the backward needs."""
    acts, a = [x], x
    for i in range(1, h + 2):
        z = a @ p["W%d" % i] + p["b%d" % i]
        a = np.maximum(z, 0.0) if i <= h else z
        acts.append(a)
    cache = dict(acts=acts)
    if "ln_scale" in p:
        ax = -1 if LN_DIMS == "row" else None
        mu = a.mean(ax, keepdims=True)
        var = ((a - mu) ** 2).mean(ax, keepdims=True)
        # y = d / D with D = sqrt(var + eps) (LN_MODE 0) or sqrt(var) + eps (LN_MODE 1); dD / dx_i = d_i / (L s), s = sqrt(var [+ eps]):
        # the xhat term of the pullback carries kappa = D / s (1 in mode 0)
        s_ = np.sqrt(var + LN_EPS) if LN_MODE == 0 else np.sqrt(var)
        den = s_ if LN_MODE == 0 else s_ + LN_EPS
        rstd = 1.0 / den
        kappa = np.where(s_ > 0, den / np.where(s_ > 0, s_, 1.0), 1.0)
        xhat = (a - mu) * rstd
        cache.update(xhat=xhat, rstd=rstd, kappa=kappa, ax=ax)
        a = xhat * p["ln_scale"] + p["ln_bias"]
    return a, cache


def _mlp_bwd(g, cache, p, h):
    """g: gradient wrt the MLP output.  Returns (gradient wrt the input, {tensor name: gradient})."""
    gp = {}
    if "ln_scale" in p:
        xhat, rstd = cache["xhat"], cache["rstd"]
        gp["ln_scale"] = (g * xhat).sum(0)
        gp["ln_bias"] = g.sum(0)
        gg = g * p["ln_scale"]
        ax = cache["ax"]
        g = rstd * (gg - gg.mean(ax, keepdims=True) - xhat * cache["kappa"] * (gg * xhat).mean(ax, keepdims=True))
    acts = cache["acts"]
    for i in range(h + 1, 0, -1):
        if i <= h:
            g = g * (acts[i] > 0.0)
        gp["W%d" % i] = acts[i - 1].T @ g
        gp["b%d" % i] = g.sum(0)
        g = g @ p["W%d" % i].T
    return g, gp


def model_vjp(packed, cfg, nf, ef, senders, receivers, seed, dtype=np.float64, set2=None):
    """Reverse pass through model(graph): seed(out) -> cotangent of out.  Returns (out, packed parameter gradient,
    cotangent of nf).  set2 = (ef2, senders2, receivers2): the second edge set of MGN-spec (forward above)."""
    h = cfg["hidden_layers"]
    P = _unpack(packed, cfg, dtype)
    if bool(cfg.get("Fe2")) != (set2 is not None):
        raise ValueError("model_vjp: cfg['Fe2'] and set2 go together")
    nf, ef = np.asarray(nf, dtype), np.asarray(ef, dtype)
    N = nf.shape[0]
    L = cfg["L"]
    sets = [("", senders, receivers)]
    v, c_en = _mlp_fwd(nf, P["enc_node"], h)
    e, c_ee = _mlp_fwd(ef, P["enc_edge"], h)
    es, c_encs = [e], [c_ee]
    if set2 is not None:
        e2, c_ee2 = _mlp_fwd(np.asarray(set2[0], dtype), P["enc_edge2"], h)
        es.append(e2)
        c_encs.append(c_ee2)
        sets.append(("2", set2[1], set2[2]))
    caches = []
    for k in range(cfg["mps"]):
        c_es, aggs, news = [], [], []
        for (sfx, s_, r_), e_ in zip(sets, es):
            e_new, c_e = _mlp_fwd(np.concatenate([v[s_], v[r_], e_], 1), P["proc%d_edge%s" % (k, sfx)], h)
            aggs.append(scatter_add(e_new, r_, N))
            c_es.append(c_e)
            news.append(e_new)
        v_new, c_v = _mlp_fwd(np.concatenate([v] + aggs, 1), P["proc%d_node" % k], h)
        caches.append((c_es, c_v))
        v = v + v_new
        es = [e_ + n_ for e_, n_ in zip(es, news)]
    out, c_d = _mlp_fwd(v, P["decoder"], h)

    G = {}
    gv, G["decoder"] = _mlp_bwd(seed(out), c_d, P["decoder"], h)
    ges = [np.zeros_like(e_) for e_ in es]
    for k in range(cfg["mps"] - 1, -1, -1):
        c_es, c_v = caches[k]
        g_in, G["proc%d_node" % k] = _mlp_bwd(gv, c_v, P["proc%d_node" % k], h)     # v_{k+1} = v_k + MLP_v([v_k; agg_1; agg_2])
        gv = gv + g_in[:, :L]
        for q, (sfx, s_, r_) in enumerate(sets):
            g_enew = ges[q] + g_in[:, (1 + q) * L:(2 + q) * L][r_]                 # e' feeds e_{k+1} and agg[receiver]
            g_cat, G["proc%d_edge%s" % (k, sfx)] = _mlp_bwd(g_enew, c_es[q], P["proc%d_edge%s" % (k, sfx)], h)
            np.add.at(gv, s_, g_cat[:, :L])
            np.add.at(gv, r_, g_cat[:, L:2 * L])
            ges[q] = ges[q] + g_cat[:, 2 * L:]
    g_nf, G["enc_node"] = _mlp_bwd(gv, c_en, P["enc_node"], h)
    for q, (sfx, _, _) in enumerate(sets):
        _, G["enc_edge%s" % sfx] = _mlp_bwd(ges[q], c_encs[q], P["enc_edge%s" % sfx], h)
    chunks = []
    for bname, tensors in model_layout(cfg["Fn"], cfg["Fe"], cfg["O"], cfg["L"], h, cfg["mps"], cfg.get("Fe2")):
        for tname, shape in tensors:
            assert G[bname][tname].shape == tuple(shape)
            chunks.append(G[bname][tname].ravel())
    return out, np.concatenate(chunks), g_nf


def step_grads(packed, cfg, nf, ef, senders, receivers, target, mask, set2=None):
    """(gs, loss) of step!: loss = mean(mse_reduce(target, model(graph))[mask]); gs = d loss / d ps in packed order.
    mask: integer node indices (0-based here; Int32 1-based at the Julia boundary, src/MeshGraphNets.jl:352)."""
    target = np.asarray(target, np.float64)
    mask = np.asarray(mask).reshape(-1)

    def seed(out):
        g_out = np.zeros_like(out)
        np.add.at(g_out, mask, 2.0 * (out[mask] - target[mask]) / mask.size)
        return g_out

    out, gs, _ = model_vjp(packed, cfg, nf, ef, senders, receivers, seed, set2=set2)
    return gs, float(mse_reduce(target, out)[mask].mean())


def ode_vjp(packed, cfg, x, node_type_onehot, ef_raw, senders, receivers, n_norm_fields, n_norm_type, e_norm, o_norm, val_mask, lam,
            set2=None):
    """Vector-Jacobian product of the RHS f = ode_rhs (no inflow overwrite) for solver-based training
    (src/strategies.jl:175-196): returns (lambda^T df/dx, lambda^T df/dps, f(x)).  Normalisers must be affine
    (frozen), given as objects with .affine(dim) -> (scale, shift)."""
    x = np.asarray(x, np.float64)
    O = x.shape[1]
    nf = np.concatenate([n_norm_fields(x), n_norm_type(node_type_onehot)], 1)
    ef = e_norm(ef_raw)
    vm = np.asarray(val_mask, np.float64).reshape(-1, 1)
    o_scale = np.broadcast_to(o_norm.std, (O,))            # inverse_data: out * std + mean
    out, gs, g_nf = model_vjp(packed, cfg, nf, ef, senders, receivers, lambda out: np.asarray(lam, np.float64) * vm * o_scale, set2=set2)
    n_scale, _ = n_norm_fields.affine(O)
    return g_nf[:, :O] * n_scale, gs, o_norm.inverse(out) * vm


def loss_only(packed, cfg, nf, ef, senders, receivers, target, mask, set2=None):
    out = forward(packed, cfg, nf, ef, senders, receivers, set2=set2)
    return float(mse_reduce(target, out)[np.asarray(mask).reshape(-1)].mean())


# --------------------------------------------------------------------------------------------
# ODE right-hand side wrapper  (src/solve.jl:147-158 ode_func_eval, :188-219 ode_step,
#                               src/graph.jl:75-97 build_graph)
# --------------------------------------------------------------------------------------------
def ode_rhs(packed, cfg, x, node_type_onehot, ef_raw, senders, receivers, n_norm_fields, n_norm_type,
            e_norm, o_norm, val_mask, inflow_mask=None, inflow_values=None):
    """x: [N][O] state (target fields stacked).  Returns dx/dt [N][O].
    - inflow overwrite x[inflow_mask] = gt[inflow_mask]          (src/solve.jl:151-152)
    - nf = [n_norm[field](x) ..., n_norm['node_type'](onehot)]    (src/graph.jl:80-86)
    - ef = e_norm(edge_features)                                 (src/graph.jl:93)
    - out = model(graph); inverse_data(o_norm, out) .* val_mask  (src/solve.jl:200-218)"""
    x = np.array(x, np.float64)
    if inflow_mask is not None:
        x[inflow_mask] = np.asarray(inflow_values, np.float64)[inflow_mask]
    nf = np.concatenate([n_norm_fields(x), n_norm_type(node_type_onehot)], 1)
    ef = e_norm(ef_raw)
    out = forward(packed, cfg, nf, ef, senders, receivers)
    return o_norm.inverse(out) * val_mask


def inflow_frame(t, saves_dt, rule="reference", time_type=np.float64):
    """0-based frame of the inflow data a right-hand side at time t reads: the reference's `floor(Int, t / saves_dt) + 1`
    (src/solve.jl:151, 1-based there) with the quotient formed in the solver's time type and NO tolerance -- in Float64
    0.29 / 0.01 = 28.999999999999996 floors to 28, so a step whose time sits an ulp below a frame boundary re-uses the previous
    frame; this is what the reference computes and it is kept.  rule = "tolerant": floor(t / saves_dt + 1e-3) (step k reads frame k
    whatever the time type: an accumulated Float32 time drifts by ~1e-4 frames)."""
    T = np.dtype(time_type).type
    if rule == "tolerant":
        return int(np.floor(float(t) / float(saves_dt) + 1e-3))
    return int(np.floor(T(T(t) / T(saves_dt))))


def euler_times(t0, dt, nsteps, time_type=np.float64):
    """The times a fixed-step integrator hands to its right-hand side: t <- t + dt in its time type, step after step
    (OrdinaryDiffEq's loop footer adds dt to t and snaps only to tstops; `solve(...; adaptive = false, dt, saveat)` at
    src/solve.jl:60 has none but the end of the interval)  [ODE-unverified: DifferentialEquations.jl cannot be run here]."""
    T = np.dtype(time_type).type
    ts, t = [], T(t0)
    for _ in range(nsteps):
        ts.append(t)
        t = T(t + T(dt))
    return ts


def euler_rollout(rhs, x0, dt, nsteps, inflow_mask=None, inflow_values=None, rule=None, time_type=np.float64, saves_dt=None):
    """Fixed-step Euler (solve(prob, Euler(); adaptive=false, dt) at src/solve.jl:60).
    Reference quirk kept on purpose: ode_func_eval overwrites the inflow rows of the array the solver hands it
    IN PLACE (`x[inflow_mask] = ...`, src/solve.jl:151-152), and for an out-of-place Euler step that array is
    the integrator's own state, so the overwrite is part of the state the step starts from:
        x_k' = overwrite(x_k);  x_{k+1} = x_k' + dt * f(x_k').   Saved values are the x_k BEFORE the overwrite
    of step k (saveat stores u after the step).
    rule = None: frame = step number (the fixtures GOLD-D / GOLD-E; equal to the tolerant rule); "reference" / "tolerant":
    inflow_frame(t_k, saves_dt or dt, rule, time_type) on the integrator's own times (euler_times); rhs receives that t_k."""
    xs = [np.array(x0, np.float64)]
    ts = euler_times(0.0, dt, nsteps, time_type) if rule is not None else [i * dt for i in range(nsteps)]
    for i in range(nsteps):
        xk = xs[-1].copy()
        if inflow_mask is not None:
            fr = i if rule is None else inflow_frame(ts[i], dt if saves_dt is None else saves_dt, rule, time_type)
            xk[inflow_mask] = np.asarray(inflow_values[fr], np.float64)[inflow_mask]    # IndexError == the reference's BoundsError
        xs.append(xk + dt * rhs(xk, ts[i]))
    return np.stack(xs)


# --------------------------------------------------------------------------------------------
# Training-step forward part  (step!(mgn, graph, target, mask, mse_reduce), src/strategies.jl:421)
# --------------------------------------------------------------------------------------------
def mse_reduce(target, out):
    return ((target - out) ** 2).sum(-1)


def step_loss(packed, cfg, nf, ef, senders, receivers, target, mask_idx):
    out = forward(packed, cfg, nf, ef, senders, receivers)
    return mse_reduce(target, out)[mask_idx].mean()


# --------------------------------------------------------------------------------------------
# Tsit5 with PI step control and tstops = saveat (the adaptive branch of rollout, src/solve.jl:58-59).
# DifferentialEquations.jl itself cannot be run here: this restates the published Tsitouras 5(4) pair and the
# standard PI controller (beta1 = 7/50, beta2 = 2/25, gamma = 0.9, qmin = 0.2, qmax = 10) used as its default.
# --------------------------------------------------------------------------------------------
TS_C = np.array([0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0])
TS_A = np.zeros((7, 6))
TS_A[1, :1] = [0.161]
TS_A[2, :2] = [-0.008480655492356989, 0.335480655492357]
TS_A[3, :3] = [2.8971530571054935, -6.359448489975075, 4.3622954328695815]
TS_A[4, :4] = [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525]
TS_A[5, :5] = [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383]
TS_A[6, :6] = [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]
TS_BT = np.array([-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
                  0.5823571654525552, -0.45808210592918697, 0.015151515151515152])


def tsit5_rollout(f, x0, t0, t1, saves, abstol=1e-6, reltol=1e-3, dt0=0.0):
    """f(x, t) may modify x in place (the inflow overwrite).  Returns (solution at `saves`, stats)."""
    beta1, beta2, gamma, qmin, qmax = 7 / 50, 2 / 25, 0.9, 0.2, 10.0

    def nrm(v, a, b):
        return float(np.sqrt(np.mean((v / (abstol + reltol * np.maximum(np.abs(a), np.abs(b)))) ** 2)))

    u = np.array(x0, np.float64)
    t, qold = float(t0), 1e-4
    out, si, nacc, nrej = [], 0, 0, 0
    if abs(saves[0] - t0) < 1e-12:      # the initial value is saved BEFORE the first f call can overwrite inflow rows
        out.append(u.copy())            # (OrdinaryDiffEq saves u0 in init, then initialises fsalfirst = f(u0))
        si = 1
    k = [None] * 7
    k[0] = f(u, t)
    nrhs = 1
    dt = dt0
    if dt <= 0:
        d0, d1 = nrm(u, u, u), nrm(k[0], u, u)
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        ut = u + h0 * k[0]
        f1 = f(ut, t + h0)
        nrhs += 1
        d2 = nrm((f1 - k[0]) / h0, u, u)
        mx = max(d1, d2)
        h1 = max(1e-6, h0 * 1e-3) if mx <= 1e-15 else (0.01 / mx) ** 0.2
        dt = min(100 * h0, h1)
    while t < t1 - 1e-12 * abs(t1) - 1e-15:
        tstop = min(saves[si] if si < len(saves) else t1, t1)
        h, hit = dt, False
        if t + h >= tstop - 1e-9 * abs(tstop):
            h, hit = tstop - t, True
        for s_ in range(1, 7):
            us = u + h * sum(TS_A[s_, j] * k[j] for j in range(s_))
            k[s_] = f(us, t + TS_C[s_] * h)
            nrhs += 1
        unew = us
        est = nrm(h * sum(TS_BT[j] * k[j] for j in range(7)), u, unew)
        q11 = max(est, 1e-30) ** beta1
        if est <= 1.0:
            q = max(1 / qmax, min(1 / qmin, q11 / qold ** beta2 / gamma))
            qold = max(est, 1e-4)
            u, k[0] = unew, k[6]
            t = tstop if hit else t + h
            nacc += 1
            dt = h / q if (not hit or h >= dt * (1 - 1e-9)) else max(dt, h / q)
            if hit and si < len(saves) and abs(saves[si] - t) <= 1e-9 * abs(t) + 1e-12:
                out.append(u.copy())
                si += 1
        else:
            nrej += 1
            dt = h / min(1 / qmin, q11 / gamma)
    while len(out) < len(saves):
        out.append(u.copy())
    return np.stack(out), dict(n_accept=nacc, n_reject=nrej, n_rhs=nrhs)
