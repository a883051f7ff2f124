// mgn_step: the training step GraphNetCore.step!(mgn, graph, target, mask, mse_reduce) of the reference
// (src/strategies.jl:418-422; gradients applied at src/MeshGraphNets.jl:370-378) behind the C ABI.
// Host orchestration only -- weights repacked into training order (forward and transposed chunks), an arena of kept
// activations, kernel sequencing; all arithmetic is in train.hip.  No CPU compute path.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "engine_internal.h"
#include "train.h"

namespace mgn {

namespace {

// The training kernels (train.hip) run ONE shape: Dense - ReLU - Dense - ReLU - Dense (+ LayerNorm, + residual), the reference's
// default hidden_layers = 2.  Other depths are expressed in that shape with identity slots, exactly (an identity product of an
// fp32 row is that row; ReLU of a value that is already a ReLU output is that value, and its mask in the reverse pass is the
// same mask):
//   hidden_layers = 1:  [D0, I, D1]
//   hidden_layers = 3:  [D0, I, I] -> [D1, D2, D3]        (the first block's output is ReLU(z0): no LayerNorm, no residual)
//   hidden_layers = 4:  [D0, D1, I] -> [D2, D3, D4]
// One such launch unit with its device offsets (floats into TrainState::w) in training order:
constexpr int DEFER_PB_JOBS = 12;  // column-sum slots per launch unit of the deferred reduction (a unit has at most 3 + 5 + 2)
constexpr int LNSUM_GROUPS = 64;   // first-level groups of the in-kernel LayerNorm-parameter sums

struct TrainBlock {
    int nin = 1;                 // L-wide layer-1 input blocks
    int in_rows = 0;             // rows of W1 that exist (< L for the encoders: zero-padded chunk)
    size_t W1[3] = {0, 0, 0}, W2 = 0, W3 = 0, W2T = 0, W3T = 0, tabs = 0;
    size_t W1T[3] = {0, 0, 0};
    bool has_w1t = false, ln = false;
    long gW[3] = {-1, -1, -1}, gb[3] = {-1, -1, -1};   // packed-parameter offsets of the three slots (-1: identity slot, nothing to learn)
    long ggamma = -1, gbeta = -1;
    int out_cols = 0;            // columns of slot 3 that exist (decoder: O)
};
struct TrainMlp {
    int nblk = 1;
    TrainBlock b[2];
    int lnslot = -1;             // whole-array LayerNorm (ln_dims = MGN_LN_ALL): which (mean, rden, kappa) slot of the arena belongs to this MLP
};
// kept activations of one MLP instance: per block H1, H2, Y (arena offsets)
struct Acts { size_t h[2][3] = {{0, 0, 0}, {0, 0, 0}}; };

}  // namespace

struct TrainState {
    bool packed = false, graph_ready = false;
    bool factored[MAX_EDGE_SETS] = {false, false};   // edge MLPs with a factored first layer (P = v W1s, Q = v W1r per node): large meshes
    bool recompute = false;      // some processor steps keep only their inputs; their H1 / H2 / Y are recomputed in the reverse pass
    int keep_steps = 0;          // the LAST keep_steps processor steps store H1 / H2 / Y (all of them when !recompute)
    bool kept(int k, int mps) const { return k >= mps - keep_steps; }
    int nblk = 1;                // launch units per MLP (2 for hidden_layers 3, 4)
    DevBuf w;                    // training-order weights
    DevBuf pk_params, pk_tabs, pk_jobs, pk_max;   // what k_pack_train builds them from: the parameter vector, the packed tables, the chunk list
    // enc-node, enc-edge (per set), per step: edge (per set), node; decoder
    TrainMlp m_en, m_de, m_ee[MAX_EDGE_SETS];
    std::vector<TrainMlp> m_pe[MAX_EDGE_SETS], m_pn;
    DevBuf arena, idx, grads, target, mask, loss;
    std::vector<int32_t> g2l, mask_host;       // renumbered graph: caller's node id -> engine row; the mapped mask of the call
    std::vector<int32_t> mask_seen;            // the (host) mask the device copy was made from, as the caller gave it
    bool mask_valid = false;
    int32_t mask_base = 0;
    // whole-array LayerNorm mode (lnall_*): its own small arena (no kept activations), the edge gids as int32
    DevBuf la, la_idx;
    bool la_ready = false;
    size_t arena_floats = 0;
    // arena offsets (floats)
    size_t nf_raw, nf_pad, ef_raw[MAX_EDGE_SETS], ef_pad[MAX_EDGE_SETS], V0, Enew;
    Acts a_en, a_de, a_ee[MAX_EDGE_SETS];
    std::vector<Acts> a_pe[MAX_EDGE_SETS], a_pn;
    std::vector<size_t> Ek[MAX_EDGE_SETS], Vk, agg[MAX_EDGE_SETS];
    static constexpr int GSETS_MAX = 72;
    static inline int GSETS = [] { const char* e = getenv("MGN_TRAIN_GSETS"); const int v = e ? atoi(e) : 4; return v < 2 ? 2 : (v > 72 ? 72 : v); }();   // gradient-buffer sets: the weight gradients of unit i run beside the backward of units i+1 .. i+3
    int gsets = 1;                    // sets allocated for the current graph (GSETS on small meshes, else 1: no overlap)
    size_t GT[GSETS_MAX], GXH[GSETS_MAX], GY[GSETS_MAX], GZ2[GSETS_MAX], GZ1[GSETS_MAX];
    size_t GXs, GXr, GXB, gV[2], gE[MAX_EDGE_SETS][2], gAgg[MAX_EDGE_SETS], Gout, gNF, io, ptmp, pw, pb;
    size_t Pn, Qn, SGs, SGr;     // factored first layer: per-node projections (forward) and summed GZ1 rows (backward)
    // whole-array LayerNorm in the training step: per-MLP statistics of the forward (64 floats each), the double partials of the two
    // reductions, (m1, m2) of the pullback
    size_t lnstats = 0, lnpart = 0, lnm = 0;
    bool defer_reduce = false;        // small meshes (second stream): every unit's partial sums are kept and reduced in a few launches at the end of the reverse pass
    size_t pw_all = 0, pb_all = 0;    // ... their partial blocks: [unit][5][nb][L][L], [unit][DEFER_PB_JOBS][nb][L]
    int defer_units = 0;
    bool need_gt = true;              // GT / GXH are full arrays (else 64-float stubs: every LayerNorm'd unit runs with LNSUM)
    size_t lnsum = 0, lnsum2 = 0;     // per-block LayerNorm-parameter sums of the streaming backward kernel and their first-level reduction (TrainBwdArgs::LNSUM)
    size_t segcarry = 0;              // carry rows of the fused aggregation (2 per edge tile; TrainFwdArgs::SEG_CARRY)
    size_t lnrow = 0;                 // (mean, 1 / denominator) per row of the MLP being unwound (TrainBwdArgs::LNROW)
    // weight gradients + their reductions go to a second stream (small meshes leave most of the chip idle during k_mlp_bwd)
    hipStream_t aux = nullptr;
    hipEvent_t ev_bwd = nullptr, ev_wg[GSETS_MAX] = {};
    // hipGraph replay of the two launch sequences over fixed buffers (small meshes): [0] forward, [1] backward of mgn_step,
    // [2] backward of mgn_ode_vjp (it also produces the input gradient).  Eager once, captured on the next call.
    hipGraphExec_t exec[3] = {nullptr, nullptr, nullptr};
    bool warm[3] = {false, false, false};
    const void* exec_arena = nullptr; const void* exec_w = nullptr;
    void drop_graphs() {
        for (int i = 0; i < 3; ++i) {
            if (exec[i]) (void)hipGraphExecDestroy(exec[i]);
            exec[i] = nullptr;
            warm[i] = false;
        }
    }
    ~TrainState() {
        drop_graphs();
        if (ev_bwd) (void)hipEventDestroy(ev_bwd);
        for (hipEvent_t e : ev_wg) if (e) (void)hipEventDestroy(e);
        if (aux) (void)hipStreamDestroy(aux);
    }
    // idx buffer (int32), per edge set: egid32 [E], perm_s [E], rowptr_s [N+1]
    size_t i_egid[MAX_EDGE_SETS] = {0, 0}, i_perm[MAX_EDGE_SETS] = {0, 0}, i_rowptr_s[MAX_EDGE_SETS] = {0, 0};
};

void train_invalidate(mgn_engine* h, int what) {
    if (!h || !h->train) return;
    if (what & 1) h->train->packed = false;
    if (what & 2) { h->train->graph_ready = false; h->train->la_ready = false; }
}

void train_free(mgn_engine* h) {
    if (!h) return;
    delete h->train;
    h->train = nullptr;
}

namespace {

int pack_training_weights(mgn_engine* h) {
    TrainState& T = *h->train;
    const mgn_config& c = h->cfg;
    const int L = c.L;
    const size_t CH = (size_t)L * L;
    const float* p = h->params.data();
    // A training loop re-packs after every optimiser update (the parameters change before every step!).  The host only describes the
    // ~300 chunk copies and packs the small tables; the parameter vector goes to the device once (9 MB instead of 42 MB of packed
    // copies) and k_pack_train writes every fragment-order / t-major / transposed / padded chunk there: 26 ms of host packing and
    // upload -> ~2 ms per update, against a 3.4 ms step on the cylinder mesh.
    size_t fsz = 0;                                   // floats of the packed buffer T.w
    std::vector<PackJob> jobs;
    std::vector<float> tabs;                          // the tables, compact (T_COUNT * L per block), scattered by the same kernel
    // rows [r0, r0 + nr) x cols [0, nc) of W (leading dimension ldw), zero-padded to L x L; transposed on request
    // L = 128: every chunk also as two fp16 pieces times a power of two that puts its largest entry into [2^14, 2^15) (split_common.hpp;
    // the streaming kernels of large launches compute on them), at + 2 CH, and 1 / that power at + 3 CH
    const bool pieces = L == 128;
    auto block = [&](const float* Wm, int ldw, int r0, int nr, int nc, bool transpose) {
        const size_t off = fsz;
        fsz += pieces ? 3 * CH + 4 : 2 * CH;          // fragment order, then the t-major copy (cooperative kernels) at + CH
        const float sc = pieces ? 1.f : 0.f;          // (> 0: pieces wanted; their scale is found on the device, k_pack_absmax)
        jobs.push_back({(long long)off, Wm ? (long long)(Wm - p) : -1LL, ldw, r0, nr, nc, transpose ? 1 : 0, 0, sc});
        return off;
    };
    const size_t ident = block(nullptr, 0, 0, 0, 0, false);   // (its own transpose)
    auto build = [&](const MlpOff& m, bool need_input_grad) {
        TrainMlp t;
        const int nd = m.nl;                          // Dense layers: hidden_layers + 1
        int plan[2][3] = {{0, 1, 2}, {-1, -1, -1}};
        t.nblk = nd <= 3 ? 1 : 2;
        if (nd == 2) { plan[0][1] = -1; plan[0][2] = 1; }
        if (nd == 4) { plan[0][1] = plan[0][2] = -1; plan[1][0] = 1; plan[1][1] = 2; plan[1][2] = 3; }
        if (nd == 5) { plan[0][2] = -1; plan[1][0] = 2; plan[1][1] = 3; plan[1][2] = 4; }
        for (int bi = 0; bi < t.nblk; ++bi) {
            TrainBlock& b = t.b[bi];
            const int d0 = plan[bi][0], d1 = plan[bi][1], d2 = plan[bi][2];
            const bool last = bi == t.nblk - 1;
            b.nin = (bi == 0 && m.in >= L) ? m.in / L : 1;
            b.in_rows = (bi == 0 && m.in < L) ? m.in : L;
            b.has_w1t = need_input_grad || bi > 0;    // (a second block always hands its input gradient to the first)
            for (int j = 0; j < b.nin; ++j) {
                b.W1[j] = block(p + m.W[d0], L, j * L, b.in_rows, L, false);
                if (b.has_w1t) b.W1T[j] = block(p + m.W[d0], L, j * L, b.in_rows, L, true);
            }
            b.gW[0] = (long)m.W[d0]; b.gb[0] = (long)m.b[d0];
            if (d1 >= 0) {
                b.W2 = block(p + m.W[d1], L, 0, L, L, false);
                b.W2T = block(p + m.W[d1], L, 0, L, L, true);
                b.gW[1] = (long)m.W[d1]; b.gb[1] = (long)m.b[d1];
            } else b.W2 = b.W2T = ident;
            b.out_cols = L;
            if (d2 >= 0) {
                b.out_cols = d2 == nd - 1 ? m.out : L;
                b.W3 = block(p + m.W[d2], b.out_cols, 0, L, b.out_cols, false);
                b.W3T = block(p + m.W[d2], b.out_cols, 0, L, b.out_cols, true);
                b.gW[2] = (long)m.W[d2]; b.gb[2] = (long)m.b[d2];
            } else b.W3 = b.W3T = ident;
            b.ln = last && m.ln;
            // tables: biases (zero behind an identity slot), LayerNorm parameters of the last block
            const size_t off = fsz, tpos = tabs.size();
            fsz += (size_t)T_COUNT * L;
            tabs.resize(tpos + (size_t)T_COUNT * L, 0.f);
            jobs.push_back({(long long)off, (long long)tpos, 0, 0, 0, 0, 0, 1, 0.f});
            float* tb_ = tabs.data() + tpos;
            std::vector<float> bias(L, 0.f);
            pack_tab(tb_ + (size_t)T_B1 * L, p + m.b[d0], L);
            if (d1 >= 0) pack_tab(tb_ + (size_t)T_B2 * L, p + m.b[d1], L);
            if (d2 >= 0) {
                for (int i = 0; i < b.out_cols; ++i) bias[i] = p[m.b[d2] + i];
                pack_tab(tb_ + (size_t)T_B3 * L, bias.data(), L);
            }
            if (b.ln) {
                pack_tab(tb_ + (size_t)T_GAMMA * L, p + m.gamma, L);
                pack_tab(tb_ + (size_t)T_BETA * L, p + m.beta, L);
                b.ggamma = (long)m.gamma; b.gbeta = (long)m.beta;
            }
            tb_[(size_t)T_LN * L] = c.ln_mode == MGN_LN_STD_EPS ? 0.f : 1e-5f;         // (eps_in, eps_out) of the LayerNorm variant
            tb_[(size_t)T_LN * L + 1] = c.ln_mode == MGN_LN_STD_EPS ? 1e-5f : 0.f;     // (frag.hpp: ln_rstd_at; both are trained)
            b.tabs = off;
        }
        return t;
    };
    T.nblk = c.hidden_layers >= 3 ? 2 : 1;
    T.m_en = build(h->enc_node, true);                // input gradient: mgn_ode_vjp (d f / d x)
    T.m_pn.clear();
    for (int q = 0; q < h->nsets; ++q) {
        T.m_ee[q] = build(h->es[q].enc, false);
        T.m_pe[q].clear();
        for (int k = 0; k < c.mps; ++k) T.m_pe[q].push_back(build(h->es[q].pe[k], true));
    }
    for (int k = 0; k < c.mps; ++k) T.m_pn.push_back(build(h->pn[k], true));
    T.m_de = build(h->dec, true);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, T.w.ensure(fsz * 4));
    HIPCHK(h, T.pk_params.ensure(h->params.size() * 4));
    HIPCHK(h, T.pk_tabs.ensure(tabs.size() * 4));
    HIPCHK(h, T.pk_jobs.ensure(jobs.size() * sizeof(PackJob)));
    HIPCHK(h, T.pk_max.ensure(jobs.size() * sizeof(unsigned)));
    HIPCHK(h, hipMemcpyAsync(T.pk_params.p, p, h->params.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(T.pk_tabs.p, tabs.data(), tabs.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(T.pk_jobs.p, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_pack_train(L, T.pk_jobs.as<PackJob>(), (int)jobs.size(), T.pk_params.as<float>(), T.pk_tabs.as<float>(), T.pk_max.as<unsigned>(), T.w.as<float>(), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));       // (jobs / tabs are locals: their copies must have left the host)
    HIPCHK(h, T.grads.ensure(h->params.size() * 4));
    T.packed = true;
    return MGN_OK;
}

int prepare_graph(mgn_engine* h) {
    TrainState& T = *h->train;
    const LocalGraph& g = h->g;
    const int L = h->cfg.L, mps = h->cfg.mps, S = h->nsets, NB = T.nblk;
    const int64_t N = g.n_own;
    const size_t NL = (size_t)(N > 0 ? N : 1) * L;
    size_t EL[MAX_EDGE_SETS] = {0, 0}, ELmax = 0;
    int64_t Emax = 0;
    // index arrays per set: edge_gid as int32, sender CSR over the receiver-sorted edge list
    std::vector<int32_t> ix;
    for (int q = 0; q < S; ++q) {
        const EdgeTopo& t = g.set[q];
        const int64_t E = t.e_local;
        if ((int64_t)t.snd.size() != E || (int64_t)t.edge_gid.size() != E)
            return fail(h, MGN_E_STATE, "the training step needs the host copy of the edge lists: install the graph with mgn_set_graph / mgn_set_edge_set");
        EL[q] = (size_t)(E > 0 ? E : 1) * L;
        ELmax = std::max(ELmax, EL[q]);
        Emax = std::max(Emax, E);
        const size_t base = ix.size();
        ix.resize(base + (size_t)2 * E + N + 1, 0);
        T.i_egid[q] = base;
        T.i_perm[q] = base + E;
        T.i_rowptr_s[q] = base + 2 * E;
        for (int64_t i = 0; i < E; ++i) ix[T.i_egid[q] + i] = (int32_t)t.edge_gid[i];
        int32_t* rp = ix.data() + T.i_rowptr_s[q];
        for (int64_t i = 0; i < E; ++i) ++rp[t.snd[i] + 1];
        for (int64_t n = 0; n < N; ++n) rp[n + 1] += rp[n];
        std::vector<int32_t> cur(rp, rp + N);
        for (int64_t i = 0; i < E; ++i) ix[T.i_perm[q] + cur[t.snd[i]]++] = (int32_t)i;   // stable: ascending edge position
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, T.idx.ensure(ix.size() * 4));
    HIPCHK(h, hipMemcpy(T.idx.p, ix.data(), ix.size() * 4, hipMemcpyHostToDevice));

    auto layout = [&](int keep) -> size_t {
        size_t off = 0;
        auto take = [&](size_t n) { const size_t o = off; off += (n + 63) / 64 * 64; return o; };
        auto take_acts = [&](size_t n) { Acts a; for (int b = 0; b < NB; ++b) for (int i = 0; i < 3; ++i) a.h[b][i] = take(n); return a; };
        T.nf_raw = take((size_t)N * h->cfg.Fn);
        T.nf_pad = take(NL);
        T.a_en = take_acts(NL);
        T.a_de = take_acts(NL);
        for (int q = 0; q < S; ++q) {
            T.ef_raw[q] = take((size_t)g.set[q].e_local * h->es[q].Fe);
            T.ef_pad[q] = take(EL[q]);
            T.a_ee[q] = take_acts(EL[q]);
        }
        // e' = LayerNorm(Y) as an array of its own only where some edge set's forward does not aggregate inside its launch (train.h: SEG_*)
        bool need_enew = h->cfg.ln_dims == MGN_LN_ALL;
        for (int q = 0; q < S; ++q) {
            const int64_t E = g.set[q].e_local;
            need_enew = need_enew || !(E > 0 && train_fwd_fused_agg(L, (int)((E + TILE - 1) / TILE)));
        }
        T.Enew = take(need_enew ? ELmax : 64);
        T.Vk.assign(mps + 1, 0);
        T.Vk[0] = take(NL);
        for (int q = 0; q < S; ++q) {
            T.Ek[q].assign(mps + 1, 0);
            T.agg[q].assign(mps, 0);
            T.a_pe[q].assign(mps, Acts());
            T.Ek[q][0] = take(EL[q]);
        }
        T.a_pn.assign(mps, Acts());
        // Kept activations of the processor: 3 (E + N) L floats per launch unit and step when H1 / H2 / Y are stored.  Small meshes store
        // them all.  Large ones store them for as many steps as the device's free memory holds (the last ones: the reverse pass meets them
        // first) and recompute the rest in the reverse pass (one more forward per MLP; (E + 2 N) L floats per step kept): M-1M needs
        // 61 GB with every step recomputed and 10.7 GB more per stored step, which is worth 4 ms of the step -- the part has 288 GB.
        // The default takes what hipMemGetInfo reports free minus a reserve (MGN_TRAIN_RESERVE_GB, 16): the FIRST handle of a process (or the
        // first process on a device) gets the stored steps, a later one sees what is left and recomputes; the size of the arena with no
        // step stored is taken from a dry run of this very layout, and an allocation that fails all the same (another process took the
        // memory between the query and the request) is retried with fewer stored steps, down to none (train_graph_build's caller).
        // MGN_TRAIN_RECOMPUTE = 0 / 1 forces all / none, MGN_TRAIN_KEEP_STEPS = n the count.
        T.keep_steps = keep;
        T.recompute = T.keep_steps < mps;
        Acts shared_e[MAX_EDGE_SETS], shared_n;
        if (T.recompute) {
            for (int q = 0; q < S; ++q) shared_e[q] = take_acts(EL[q]);
            shared_n = take_acts(NL);
        }
        for (int k = 0; k < mps; ++k) {
            for (int q = 0; q < S; ++q) {
                T.a_pe[q][k] = T.kept(k, mps) ? take_acts(EL[q]) : shared_e[q];
                T.agg[q][k] = take(NL);
                T.Ek[q][k + 1] = take(EL[q]);
            }
            T.a_pn[k] = T.kept(k, mps) ? take_acts(NL) : shared_n;
            T.Vk[k + 1] = take(NL);
        }
        const size_t ML = NL > ELmax ? NL : ELmax;
        // Above the cooperative range the first layer of the edge MLPs is factored as in the inference kernels: per NODE
        // P = v W1_sender, Q = v W1_receiver (2 chunk passes over N rows instead of 2 over E rows), backward and weight gradients
        // through the summed rows of GZ1 (gather <-> segmented-sum duality).  MGN_TRAIN_FACTORED = 0 / 1 overrides the size rule.
        bool any_fact = false, all_fact = true;
        for (int q = 0; q < S; ++q) {
            const int64_t E = g.set[q].e_local;
            T.factored[q] = !train_uses_coop(128, (int)((E + TILE - 1) / TILE));   // the size rule of the cooperative kernels, for every L
            if (const char* e = getenv("MGN_TRAIN_FACTORED")) T.factored[q] = atoi(e) != 0;
            if (E == 0) T.factored[q] = false;
            any_fact = any_fact || T.factored[q];
            all_fact = all_fact && T.factored[q];
        }
        // Small meshes (the cooperative-tile regime: a launch leaves most of the chip idle) get GSETS sets of gradient buffers so
        // that the parameter gradients can run on a second stream; larger ones fill the chip on their own and keep one set.
        {
            static const bool overlap_env = [] { const char* e = getenv("MGN_TRAIN_OVERLAP"); return !e || atoi(e) != 0; }();
            const int64_t big = Emax > N ? Emax : N;
            T.gsets = (overlap_env && !T.recompute && !any_fact && L == 128 && big <= 2048 * TILE) ? TrainState::GSETS : 1;   // (SGs / SGr are single buffers)
        }
        // GT / G xhat rows only where some LayerNorm'd launch unit does not take its parameter sums inside the backward kernel (train.h: LNSUM)
        bool need_gt = h->cfg.ln_dims == MGN_LN_ALL || T.gsets > 1 || !train_bwd_ln_sums(L, (int)((N + TILE - 1) / TILE));
        for (int q = 0; q < S; ++q) {
            const int64_t E = g.set[q].e_local;
            need_gt = need_gt || (E > 0 && !train_bwd_ln_sums(L, (int)((E + TILE - 1) / TILE)));
        }
        T.need_gt = need_gt;
        for (int i = 0; i < T.gsets; ++i) { T.GT[i] = take(need_gt ? ML : 64); T.GXH[i] = take(need_gt ? ML : 64); T.GY[i] = take(ML); T.GZ2[i] = take(ML); T.GZ1[i] = take(ML); }
        T.GXs = T.GXr = T.Pn = T.Qn = T.SGs = T.SGr = T.GXB = 0;
        if (any_fact) { T.Pn = take(NL); T.Qn = take(NL); T.SGs = take(NL); T.SGr = take(NL); }
        if (!all_fact) { T.GXs = take(ELmax); T.GXr = take(ELmax); }
        if (NB > 1) T.GXB = take(ML);                      // gradient handed from an MLP's second launch unit to its first
        T.gV[0] = take(NL); T.gV[1] = take(NL);
        for (int q = 0; q < S; ++q) { T.gE[q][0] = take(EL[q]); T.gE[q][1] = take(EL[q]); T.gAgg[q] = take(NL); }
        T.Gout = take(NL);
        T.gNF = take(NL);
        T.io = take((size_t)(N > 0 ? N : 1) * (2 * h->cfg.O + h->cfg.Fn + 1));
        T.ptmp = take((size_t)(N > 0 ? N : 1) * (size_t)std::max(h->cfg.Fn, h->cfg.O));      // row permutations of a renumbered graph
        const int nb = std::max(wgrad_blocks(N), wgrad_blocks(Emax));
        T.pw = take((size_t)5 * (T.gsets > 1 ? T.gsets / 2 : 1) * (nb > 0 ? nb : 1) * L * L);   // one partial-dW region per weight-gradient job of a launch (a group of units on small meshes)
        T.pb = take((size_t)(WGRAD_MAX_JOBS + 1) * (nb > 0 ? nb : 1) * L);   // (+ 1: the second output of a LayerNorm job)
        {   // Deferred reductions (MGN_TRAIN_DEFER_REDUCE = 1; built, same bits, off) where the weight gradients run on the second stream: the 33
            // k_reduce_partials launches of a step as five at its end.
            static const int defer_env = [] { const char* e = getenv("MGN_TRAIN_DEFER_REDUCE"); return e ? atoi(e) : 0; }();   // (measured: 2.54 ms against 2.36 -- the per-unit partial blocks, re-used, stay in the caches; 680 MB of them do not)
            const int units = (2 + S + mps * (S + 1)) * NB;
            const size_t per_unit = ((size_t)5 * L * L + (size_t)DEFER_PB_JOBS * L) * (nb > 0 ? nb : 1);
            T.defer_reduce = defer_env && T.gsets > 1 && per_unit * units * 4 <= ((size_t)8 << 30);
            T.defer_units = units;
            if (T.defer_reduce) {
                T.pw_all = take((size_t)units * 5 * (nb > 0 ? nb : 1) * L * L);
                T.pb_all = take((size_t)units * DEFER_PB_JOBS * (nb > 0 ? nb : 1) * L);
            }
        }
        T.lnrow = take((size_t)2 * (ML / L));
        T.lnsum = take(((ML / L + TILE - 1) / TILE + 7) / 8 * (size_t)2 * L + 2 * L);
        T.lnsum2 = take((size_t)LNSUM_GROUPS * 2 * L);
        T.segcarry = take((size_t)2 * ((ELmax / L + TILE - 1) / TILE + 1) * L);
        if (h->cfg.ln_dims == MGN_LN_ALL) {
            int slot = 0;
            T.m_en.lnslot = slot++;
            for (int q = 0; q < S; ++q) T.m_ee[q].lnslot = slot++;
            for (int k = 0; k < mps; ++k) {
                for (int q = 0; q < S; ++q) T.m_pe[q][k].lnslot = slot++;
                T.m_pn[k].lnslot = slot++;
            }
            T.lnstats = take((size_t)64 * slot);
            const size_t nt_max = (std::max<size_t>(NL, ELmax) / L + TILE - 1) / TILE;
            T.lnpart = take((size_t)2 * std::max<size_t>({(size_t)2 * array_stats_blocks(), (size_t)2 * 128 * lnall_bwd_blocks(), (size_t)2 * 4 * nt_max}));
            T.lnm = take(64);
        }
        return off;
    };
    int keep0 = mps;
    {
        double rows = (double)NL;
        for (int q = 0; q < S; ++q) rows += (double)EL[q];
        const double per_step = 3.0 * NB * rows * 4.0, stored = (double)mps * per_step;
        if (stored > 48e9) {
            keep0 = 0;
            T.arena.release();                             // (an earlier graph's arena must not count as taken)
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                double reserve = 16e9;
                if (const char* e = getenv("MGN_TRAIN_RESERVE_GB")) reserve = atof(e) * 1e9;
                const double base = (double)layout(0) * 4.0;          // the arena with every step recomputed: a dry run of the layout below
                const double room = (double)free_b - base - reserve;
                if (room > 0) keep0 = (int)std::min<double>((double)mps, room / per_step);
            }
        }
        if (const char* e = getenv("MGN_TRAIN_RECOMPUTE")) keep0 = atoi(e) != 0 ? 0 : mps;
        if (const char* e = getenv("MGN_TRAIN_KEEP_STEPS")) keep0 = std::max(0, std::min(mps, atoi(e)));
    }
    T.drop_graphs();
    size_t off = 0;
    int test_fail = 0;                                    // MGN_TRAIN_TEST_FAIL_ALLOCS = n: the first n requests count as refused (tests of the retry)
    if (const char* e = getenv("MGN_TRAIN_TEST_FAIL_ALLOCS")) test_fail = atoi(e);
    for (int keep = keep0;; keep = keep / 2) {            // an allocation that fails is retried with fewer stored steps (none at last)
        off = layout(keep);
        if (test_fail-- <= 0 && T.arena.ensure(off * 4) == hipSuccess) break;
        (void)hipGetLastError();
        T.arena.release();
        if (keep == 0) return fail(h, MGN_E_OOM, "training arena: %.1f GB with every processor step recomputed do not fit the device's free memory", (double)off * 4e-9);
    }
    T.arena_floats = off;
    HIPCHK(h, T.target.ensure((size_t)(N > 0 ? N : 1) * h->cfg.O * 4));
    T.g2l.clear();
    T.mask_valid = false;
    if (h->g.renumbered) {
        T.g2l.assign((size_t)h->g.N, 0);
        for (int32_t i = 0; i < h->g.n_own; ++i) T.g2l[(size_t)h->g.own_gid[i]] = i;
    }
    T.graph_ready = true;
    return MGN_OK;
}

}  // namespace
}  // namespace mgn

using namespace mgn;

namespace {

struct TrainJob {
    bool vjp = false;
    // step!: the FeatureGraph as given, target, mask
    const float* nf = nullptr; const float* ef = nullptr; const float* target = nullptr;
    const int32_t* mask = nullptr; int64_t nmask = 0; int32_t mask_index_base = 0;
    float* loss = nullptr;
    // vjp of the RHS: raw inputs of mgn_ode_step (ef = raw edge features), cotangent lambda
    const float* x = nullptr; const float* onehot = nullptr; const float* val_mask = nullptr; const float* lambda = nullptr;
    float* dxdt = nullptr; float* xbar = nullptr;
    float* grads = nullptr;
    // vjp of the model itself (mgn_forward_vjp): nf / ef as in step!, cotangent `lambda` of the output, gradient of all of nf out
    bool fvjp = false;
    float* nfbar = nullptr; float* out = nullptr;
};

int train_prepare(mgn_handle* h, const char* who, size_t n_grads) {
    if (int rc = need(h, true, true, false, true)) return rc;    // (the training kernels pack their own weights from h->params)
    const mgn_config& c = h->cfg;
    if (c.nranks != 1) return fail(h, MGN_E_STATE, "%s drives one partition", who);
    if (c.dtype != MGN_F32) return fail(h, MGN_E_STATE, "%s computes in fp32: create the handle with dtype MGN_F32", who);
    for (int q = 1; q < h->nsets; ++q)
        if (h->g.set[q].E > 0 && !h->es[q].have_ef)
            return fail(h, MGN_E_STATE, "edge set %d has edges but no features: call mgn_set_edge_features after mgn_set_edge_set", q);
    if (n_grads != h->params.size()) return fail(h, MGN_E_ARG, "%s: grads has %zu floats, model has %zu", who, n_grads, h->params.size());
    if (!h->train) h->train = new (std::nothrow) TrainState();
    if (!h->train) return fail(h, MGN_E_OOM, "host allocation failed");
    if (!h->train->packed)
        if (int rc = pack_training_weights(h)) return rc;
    if (!h->train->graph_ready)
        if (int rc = prepare_graph(h)) return rc;
    return MGN_OK;
}

// forward with kept activations, seed, reverse pass, all parameter gradients (+ the input gradient for the VJP)
int train_run(mgn_handle* h, const TrainJob& J) {
    const mgn_config& c = h->cfg;
    TrainState& T = *h->train;
    const LocalGraph& g = h->g;
    const int S = h->nsets;
    const int64_t N = g.n_own;
    const int L = c.L, mps = c.mps, O = c.O;
    hipStream_t st = h->stream;
    float* A = T.arena.as<float>();
    const float* Wt = T.w.as<float>();
    struct SetIdx { int64_t E; int32_t nt; const int32_t *egid, *perm_s, *rowptr_s, *snd, *rcv, *rowptr; } sx[MAX_EDGE_SETS] = {};
    for (int q = 0; q < S; ++q) {
        sx[q].E = g.set[q].e_local;
        sx[q].nt = (int32_t)((sx[q].E + TILE - 1) / TILE);
        sx[q].egid = T.idx.as<int32_t>() + T.i_egid[q];
        sx[q].perm_s = T.idx.as<int32_t>() + T.i_perm[q];
        sx[q].rowptr_s = T.idx.as<int32_t>() + T.i_rowptr_s[q];
        sx[q].snd = h->es[q].d_snd.as<int32_t>();
        sx[q].rcv = h->es[q].d_rcv.as<int32_t>();
        sx[q].rowptr = h->es[q].d_rowptr.as<int32_t>();
    }
    const int32_t nt_n = (int32_t)((N + TILE - 1) / TILE);
    float* G = T.grads.as<float>();

    // ---- inputs
    const float* nrm = h->norms.as<float>();   // [node scale, shift (Fn) | edge scale, shift (Fe) | out scale, shift (O)]
    // A renumbered graph (graph_host.h: the engine's node order is not the caller's): per-node inputs are brought into the engine's
    // order as they arrive and per-node results go back through the inverse; `mask` is mapped on the host.  Edges go by edge_gid already.
    const bool renum = g.renumbered;
    const int32_t* ngid = h->d_own_gid.as<int32_t>();
    auto to_local = [&](float* buf, int width) -> hipError_t {            // buf [N][width]: caller's order -> engine's, in place
        if (!renum || width <= 0) return hipSuccess;
        if (hipError_t e = launch_permute_rows(A + T.ptmp, buf, ngid, N, width, false, st)) return e;
        return hipMemcpyAsync(buf, A + T.ptmp, (size_t)N * width * 4, hipMemcpyDeviceToDevice, st);
    };
    auto to_global = [&](float* buf, int width) -> hipError_t {           // ... and back
        if (!renum || width <= 0) return hipSuccess;
        if (hipError_t e = launch_permute_rows(A + T.ptmp, buf, ngid, N, width, true, st)) return e;
        return hipMemcpyAsync(buf, A + T.ptmp, (size_t)N * width * 4, hipMemcpyDeviceToDevice, st);
    };
    // An array the caller keeps on the device is read where it is; a host array is staged first.  Gather (renumbered graph) and padding
    // run in the one kernel that reads it (as copy + permute + copy back + pad the eager prologue of a step was 14 launches, 0.24 ms on
    // the cylinder mesh).
    auto on_device = [&](const void* ptr) {
        hipPointerAttribute_t at{};
        const bool dev = ptr && hipPointerGetAttributes(&at, ptr) == hipSuccess && (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged);
        (void)hipGetLastError();
        return dev;
    };
    auto staged = [&](const float* user, float* stage, size_t floats, const float*& out) -> hipError_t {
        if (on_device(user)) { out = user; return hipSuccess; }
        out = stage;
        return hipMemcpyAsync(stage, user, floats * 4, hipMemcpyHostToDevice, st);
    };
    if (!J.vjp || J.fvjp) {
        const float* src = nullptr;
        HIPCHK(h, staged(J.nf, A + T.nf_raw, (size_t)N * c.Fn, src));
        HIPCHK(h, launch_affine_pad_gather(src, c.Fn, nullptr, 0, nullptr, nullptr, renum ? ngid : nullptr, A + T.nf_pad, L, N, st));
        if (sx[0].E > 0) {
            HIPCHK(h, staged(J.ef, A + T.ef_raw[0], (size_t)sx[0].E * c.Fe, src));
            HIPCHK(h, launch_affine_pad(src, c.Fe, nullptr, 0, nullptr, nullptr, A + T.ef_pad[0], L, sx[0].E, st));
        }
        if (J.fvjp) {
            HIPCHK(h, hipMemcpyAsync(A + T.io + (size_t)N * O, J.lambda, (size_t)N * O * 4, hipMemcpyDefault, st));
            HIPCHK(h, to_local(A + T.io + (size_t)N * O, O));
        } else {
            if (renum) {
                HIPCHK(h, staged(J.target, A + T.ptmp, (size_t)N * O, src));
                HIPCHK(h, launch_permute_rows(T.target.as<float>(), src, ngid, N, O, false, st));
            } else {
                HIPCHK(h, hipMemcpyAsync(T.target.p, J.target, (size_t)N * O * 4, hipMemcpyDefault, st));
            }
            // the mask of the previous call is usually this call's (one trajectory, one mask: reference src/MeshGraphNets.jl:352): uploaded once
            const bool mask_dev = on_device(J.mask);
            const bool same = !mask_dev && T.mask_valid && T.mask_base == J.mask_index_base && (int64_t)T.mask_seen.size() == J.nmask &&
                              (J.nmask == 0 || memcmp(T.mask_seen.data(), J.mask, (size_t)J.nmask * 4) == 0);
            if (!same) {
                HIPCHK(h, T.mask.ensure((size_t)J.nmask * 4));
                T.mask_valid = false;
                if (renum) {                       // the caller's node ids -> engine rows (0-based from here on)
                    std::vector<int32_t> host_mask;
                    const int32_t* mk = J.mask;
                    if (mask_dev) {
                        host_mask.resize((size_t)J.nmask);
                        HIPCHK(h, hipMemcpy(host_mask.data(), J.mask, (size_t)J.nmask * 4, hipMemcpyDeviceToHost));
                        mk = host_mask.data();
                    }
                    T.mask_host.resize((size_t)J.nmask);
                    for (int64_t i = 0; i < J.nmask; ++i) T.mask_host[(size_t)i] = T.g2l[(size_t)(mk[i] - J.mask_index_base)];
                    HIPCHK(h, hipMemcpyAsync(T.mask.p, T.mask_host.data(), (size_t)J.nmask * 4, hipMemcpyHostToDevice, st));
                } else {
                    HIPCHK(h, hipMemcpyAsync(T.mask.p, J.mask, (size_t)J.nmask * 4, hipMemcpyDefault, st));
                }
                if (!mask_dev) {
                    T.mask_seen.assign(J.mask, J.mask + J.nmask);
                    T.mask_base = J.mask_index_base;
                    T.mask_valid = true;
                }
            }
        }
    } else {
        // RHS inputs exactly as mgn_ode_step takes them: nf = [n_norm(x); n_norm(onehot)], ef = e_norm(ef_raw)
        float* io = A + T.io;                  // x [N][O] | lambda [N][O] | onehot [N][Fn-O] | val_mask [N]
        HIPCHK(h, hipMemcpyAsync(io, J.x, (size_t)N * O * 4, hipMemcpyDefault, st));
        HIPCHK(h, hipMemcpyAsync(io + (size_t)N * O, J.lambda, (size_t)N * O * 4, hipMemcpyDefault, st));
        if (c.Fn > O) HIPCHK(h, hipMemcpyAsync(io + (size_t)2 * N * O, J.onehot, (size_t)N * (c.Fn - O) * 4, hipMemcpyDefault, st));
        if (J.val_mask) HIPCHK(h, hipMemcpyAsync(io + (size_t)N * (O + c.Fn), J.val_mask, (size_t)N * 4, hipMemcpyDefault, st));
        HIPCHK(h, to_local(io, O));
        HIPCHK(h, to_local(io + (size_t)N * O, O));
        if (c.Fn > O) HIPCHK(h, to_local(io + (size_t)2 * N * O, c.Fn - O));
        if (J.val_mask) HIPCHK(h, to_local(io + (size_t)N * (O + c.Fn), 1));
        HIPCHK(h, launch_affine_pad(io, O, io + (size_t)2 * N * O, c.Fn - O, h->have_nnorm ? nrm : nullptr, h->have_nnorm ? nrm + c.Fn : nullptr,
                                    A + T.nf_pad, L, N, st));
        if (sx[0].E > 0) {
            HIPCHK(h, hipMemcpyAsync(A + T.ef_raw[0], J.ef, (size_t)sx[0].E * c.Fe * 4, hipMemcpyDefault, st));
            HIPCHK(h, launch_affine_pad(A + T.ef_raw[0], c.Fe, nullptr, 0, h->have_enorm ? nrm + 2 * c.Fn : nullptr,
                                        h->have_enorm ? nrm + 2 * c.Fn + c.Fe : nullptr, A + T.ef_pad[0], L, sx[0].E, st));
        }
    }
    // further edge sets: the features installed by mgn_set_edge_features, as given (the forward path does not normalise them either)
    for (int q = 1; q < S; ++q)
        if (sx[q].E > 0)
            HIPCHK(h, launch_affine_pad(h->es[q].d_ef.as<float>(), h->es[q].Fe, nullptr, 0, nullptr, nullptr, A + T.ef_pad[q], L, sx[q].E, st));

    // ln_dims = MGN_LN_ALL: every LayerNorm takes its statistics over the whole rows x L output of its MLP: the MLP
    // kernels run without their row-wise LayerNorm; two reductions and an elementwise pass follow them in both directions
    const bool lnall = c.ln_dims == MGN_LN_ALL;
    const float ln_eps_in = c.ln_mode == MGN_LN_STD_EPS ? 0.f : 1e-5f, ln_eps_out = c.ln_mode == MGN_LN_STD_EPS ? 1e-5f : 0.f;
    // One MLP forward = one or two launch units.  `in` carries rows / ntiles and the first unit's inputs (X, xidx, PRE, preidx);
    // w1sel >= 0: only block w1sel of W1 is applied per row (the factored edge MLP: the e block).
    // keep = false: first pass of recompute mode -- H1 / H2 / Y are regenerated right before the backward, not stored here
    auto run_fwd = [&](const TrainMlp& m, const TrainFwdArgs& in, int w1sel, const Acts& act, const float* resid, float* out, float* lnout,
                       bool keep) -> hipError_t {
        for (int bi = 0; bi < m.nblk; ++bi) {
            const TrainBlock& b = m.b[bi];
            const bool last = bi == m.nblk - 1;
            TrainFwdArgs a{};
            a.rows = in.rows; a.ntiles = in.ntiles;
            int nin = b.nin;
            if (bi == 0) {
                for (int j = 0; j < 3; ++j) { a.X[j] = in.X[j]; a.xidx[j] = in.xidx[j]; }
                for (int j = 0; j < 2; ++j) { a.PRE[j] = in.PRE[j]; a.preidx[j] = in.preidx[j]; }
                if (w1sel >= 0) { a.W1[0] = Wt + b.W1[w1sel]; nin = 1; }
                else for (int j = 0; j < b.nin; ++j) a.W1[j] = Wt + b.W1[j];
            } else {                                        // the second unit reads the first one's output (its Y: no LayerNorm, no residual)
                a.X[0] = A + act.h[bi - 1][2];
                a.W1[0] = Wt + b.W1[0];
            }
            a.W2 = Wt + b.W2; a.W3 = Wt + b.W3; a.tabs = Wt + b.tabs;
            if (keep) { a.H1 = A + act.h[bi][0]; a.H2 = A + act.h[bi][1]; a.Y = A + act.h[bi][2]; }
            if (last) { a.resid = resid; a.OUT = out; a.LNOUT = lnout; a.SEG_RCV = in.SEG_RCV; a.SEG_AGG = in.SEG_AGG; a.SEG_CARRY = in.SEG_CARRY; }
            else if (!keep) a.OUT = A + act.h[bi][2];
            a.ln = b.ln ? 1 : 0;
            const bool wide = lnall && last && b.ln;      // whole-array LayerNorm: the kernel stops at Y, statistics and apply follow
            const bool want_stats = wide && (out || lnout) && in.rows > 0;   // (the recomputation of the reverse pass asks for neither: the statistics are kept)
            if (wide) { a.ln = 0; a.resid = nullptr; a.OUT = nullptr; a.LNOUT = nullptr; a.SEG_RCV = nullptr; a.Y = A + act.h[bi][2]; }
            if (want_stats) a.STATS = reinterpret_cast<double*>(A + T.lnpart);   // the kernel leaves (sum, sum of squares) of Y per tile
            if (hipError_t e = launch_mlp_fwd(L, nin, a, st)) return e;
            if (want_stats) {
                float* stats = A + T.lnstats + (size_t)64 * m.lnslot;
                const int64_t n = in.rows * L;
                if (hipError_t e = launch_array_stats_final(a.STATS, train_fwd_stat_slots(L, in.ntiles), n, ln_eps_in, ln_eps_out, stats, st)) return e;
                if (hipError_t e = launch_ln_all_apply(a.Y, stats, Wt + b.tabs + (size_t)T_GAMMA * L, Wt + b.tabs + (size_t)T_BETA * L, resid, out,
                                                       lnout, n, L, st)) return e;
            }
        }
        return hipSuccess;
    };
    auto fwd = [&](const TrainMlp& m, int64_t rows, int32_t ntiles, const float* x0, const int32_t* i0, const float* x1, const float* x2,
                   const Acts& act, const float* resid, float* out, float* lnout, bool keep = true) {
        TrainFwdArgs a{};
        a.rows = rows; a.ntiles = ntiles;
        a.X[0] = x0; a.X[1] = x1; a.X[2] = x2;
        a.xidx[0] = i0;
        return run_fwd(m, a, -1, act, resid, out, lnout, keep);
    };
    // edge MLP of step k, set q: [v_s; v_r; e] -> MLP + LayerNorm; with the factored first layer P[s] + Q[r] + e W1e
    // segagg != null: the launch aggregates e' itself (train.h: SEG_*; the caller follows up with launch_seg_fixup)
    auto fwd_edge = [&](int q, int k, const float* resid, float* out, float* lnout, bool keep = true, float* segagg = nullptr) -> hipError_t {
        const TrainMlp& m = T.m_pe[q][k];
        TrainFwdArgs a{};
        a.rows = sx[q].E; a.ntiles = sx[q].nt;
        if (segagg) { a.SEG_RCV = sx[q].rcv; a.SEG_AGG = segagg; a.SEG_CARRY = A + T.segcarry; }
        if (!T.factored[q]) {
            a.X[0] = A + T.Vk[k]; a.xidx[0] = sx[q].snd;
            a.X[1] = A + T.Vk[k]; a.xidx[1] = sx[q].rcv;
            a.X[2] = A + T.Ek[q][k];
            return run_fwd(m, a, -1, T.a_pe[q][k], resid, out, lnout, keep);
        }
        Lin2Args p{};
        p.rows = N; p.ntiles = nt_n;
        p.X0 = A + T.Vk[k]; p.W0 = Wt + m.b[0].W1[0]; p.W1 = Wt + m.b[0].W1[1];
        p.OUT0 = A + T.Pn; p.OUT1 = A + T.Qn;
        if (hipError_t e = launch_lin2(L, p, st)) return e;
        a.X[0] = A + T.Ek[q][k];
        a.PRE[0] = A + T.Pn; a.preidx[0] = sx[q].snd; a.PRE[1] = A + T.Qn; a.preidx[1] = sx[q].rcv;
        return run_fwd(m, a, 2, T.a_pe[q][k], resid, out, lnout, keep);
    };
    auto fwd_node = [&](int k, const float* resid, float* out, bool keep = true) {
        return fwd(T.m_pn[k], N, nt_n, A + T.Vk[k], nullptr, A + T.agg[0][k], S > 1 ? A + T.agg[1][k] : nullptr, T.a_pn[k], resid, out, nullptr, keep);
    };

    // Small meshes replay both launch sequences from hipGraphs (everything they touch lives at fixed addresses in the arena;
    // the inputs, the seed of the reverse pass and the results stay outside).  The captured pointers are checked per call.
    const bool graphable = h->use_graph && !h->prof && st != nullptr && T.gsets > 1;   // the NULL stream cannot be captured
    if (T.exec_arena != T.arena.p || T.exec_w != T.w.p) {
        T.drop_graphs();
        T.exec_arena = T.arena.p;
        T.exec_w = T.w.p;
    }
    auto graphed = [&](int slot, auto&& launches) -> int {
        if (graphable && T.exec[slot]) {
            HIPCHK(h, hipGraphLaunch(T.exec[slot], st));
            return MGN_OK;
        }
        if (!graphable || !T.warm[slot]) {
            T.warm[slot] = true;
            return launches();
        }
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            h->use_graph = 0;
            return launches();
        }
        const int rc = launches();
        const hipError_t ce = hipStreamEndCapture(st, &graph);
        if (rc != MGN_OK || ce != hipSuccess || !graph || hipGraphInstantiate(&T.exec[slot], graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            T.exec[slot] = nullptr;
            h->use_graph = 0;             // eager from here on
            if (rc != MGN_OK) return rc;
            return launches();
        }
        (void)hipGraphDestroy(graph);
        HIPCHK(h, hipGraphLaunch(T.exec[slot], st));
        return MGN_OK;
    };

    // ---- forward, keeping activations
    auto forward_launches = [&]() -> int {
    HIPCHK(h, fwd(T.m_en, N, nt_n, A + T.nf_pad, nullptr, nullptr, nullptr, T.a_en, nullptr, A + T.Vk[0], nullptr));
    for (int q = 0; q < S; ++q)
        HIPCHK(h, fwd(T.m_ee[q], sx[q].E, sx[q].nt, A + T.ef_pad[q], sx[q].egid, nullptr, nullptr, T.a_ee[q], nullptr, A + T.Ek[q][0], nullptr));
    for (int k = 0; k < mps; ++k) {
        for (int q = 0; q < S; ++q) {
            if (!lnall && sx[q].E > 0 && train_fwd_fused_agg(L, sx[q].nt)) {   // aggregation inside the edge launch (large meshes)
                HIPCHK(h, fwd_edge(q, k, A + T.Ek[q][k], A + T.Ek[q][k + 1], nullptr, T.kept(k, mps), A + T.agg[q][k]));
                HIPCHK(h, launch_seg_fixup(L, sx[q].rowptr, A + T.segcarry, A + T.agg[q][k], (int32_t)N, st));
                continue;
            }
            HIPCHK(h, fwd_edge(q, k, A + T.Ek[q][k], A + T.Ek[q][k + 1], A + T.Enew, T.kept(k, mps)));
            HIPCHK(h, launch_segment_sum(L, A + T.Enew, sx[q].rowptr, nullptr, nullptr, A + T.agg[q][k], (int32_t)N, st));
        }
        HIPCHK(h, fwd_node(k, A + T.Vk[k], A + T.Vk[k + 1], T.kept(k, mps)));
    }
    HIPCHK(h, fwd(T.m_de, N, nt_n, A + T.Vk[mps], nullptr, nullptr, nullptr, T.a_de, nullptr, nullptr, nullptr));
    return MGN_OK;
    };
    if (int rc = graphed(0, forward_launches)) return rc;
    const size_t y_out = T.a_de.h[T.m_de.nblk - 1][2];        // the decoder's output (its last unit's Y)

    // ---- seed of the reverse pass
    const int nlb = J.vjp ? 0 : loss_blocks(J.nmask);
    HIPCHK(h, hipMemsetAsync(A + T.Gout, 0, (size_t)N * L * 4, st));
    if (!J.vjp) {   // loss = mean(mse_reduce(target, out)[mask]) and its gradient w.r.t. out
        HIPCHK(h, T.loss.ensure((size_t)nlb * sizeof(double)));
        HIPCHK(h, launch_loss(A + y_out, L, T.target.as<float>(), O, T.mask.as<int32_t>(), J.nmask, renum ? 0 : J.mask_index_base, A + T.Gout,
                              T.loss.as<double>(), st));
    } else if (J.fvjp) {   // the cotangent of the model's output as given
        HIPCHK(h, launch_vjp_seed(A + y_out, L, O, A + T.io + (size_t)N * O, nullptr, nullptr, nullptr, A + T.Gout,
                                  J.out ? T.target.as<float>() : nullptr, N, st));
    } else {        // dx/dt = inverse_data(o_norm, out) .* val_mask  =>  d/d out = lambda .* val_mask .* out_scale
        const float* os = h->have_onorm ? nrm + 2 * c.Fn + 2 * c.Fe : nullptr;
        const float* vm = J.val_mask ? A + T.io + (size_t)N * (O + c.Fn) : nullptr;
        HIPCHK(h, launch_vjp_seed(A + y_out, L, O, A + T.io + (size_t)N * O, vm, os, os ? os + O : nullptr, A + T.Gout,
                                  J.dxdt ? T.target.as<float>() : nullptr, N, st));
    }

    // ---- backward
    // activation backward of one launch unit + all of its parameter gradients
    // The parameter gradients of unit i (k_wgrad + k_reduce_partials: they only read what k_mlp_bwd left in gradient-buffer
    // set i % gsets and the kept activations) run on a second stream beside the activation backward of the next units.  Not in
    // recompute mode (there the kept activations are shared buffers which the next step's recomputation overwrites) and not
    // on large meshes, which fill the chip on their own (prepare_graph).  MGN_TRAIN_OVERLAP = 0 keeps everything on one stream.
    const bool overlap = T.gsets > 1;
    if (overlap && !T.aux) {
        HIPCHK(h, hipStreamCreateWithFlags(&T.aux, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&T.ev_bwd, hipEventDisableTiming));
        for (hipEvent_t& e : T.ev_wg) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    int n_bwd = 0;
    // Small meshes: the weight-gradient jobs of `group` consecutive launch units go out as ONE k_wgrad + ONE k_reduce_partials launch on the
    // second stream (two groups of gradient-buffer sets in flight).  Per unit they were 66 launches per processor step!, each a
    // cross-stream dependency both ways: the main stream's backward kernels started ~10 us apart (rocprofv3 timeline, docs/experiments.md)
    // and the second stream was the critical path.  Same partial blocks (128 rows), same order of the reduction: the same bits.
    static const int group_env = [] { const char* e = getenv("MGN_TRAIN_WG_GROUP"); return e ? atoi(e) : 1; }();
    const int group = overlap ? std::max(1, std::min(group_env, T.gsets / 2)) : 1;
    int64_t lrows_all = N;
    for (int q = 0; q < S; ++q) lrows_all = std::max<int64_t>(lrows_all, sx[q].E);
    WgradBatch pwb{};
    ReduceBatch prb{};
    int pnw = 0, punits = 0, nbatch = 0;
    int64_t plrows = 0;
    std::vector<ReduceJob> deferred;                     // (T.defer_reduce) the reductions of all units, launched behind the last weight-gradient launch
    int unit_no = 0;
    int set_batch[TrainState::GSETS_MAX];                // launch number that takes the weight gradients of the unit in each buffer set
    for (int& v : set_batch) v = -1;
    auto flush = [&]() -> int {
        if (punits == 0) return MGN_OK;
        hipStream_t wst = overlap ? T.aux : st;
        if (overlap) {
            HIPCHK(h, hipEventRecord(T.ev_bwd, st));
            HIPCHK(h, hipStreamWaitEvent(wst, T.ev_bwd, 0));
        }
        if (pwb.njobs > 0 && wgrad_blocks(plrows) > 0) {
            static const int whatif = [] { const char* e = getenv("MGN_TRAIN_WHATIF"); return e ? atoi(e) : 0; }();   // diagnostic (wrong gradients): 1 no weight-gradient launches, 2 no reductions
            if (!(whatif & 1)) HIPCHK(h, launch_wgrad(L, pwb, plrows, wst));
            if (!(whatif & 2)) HIPCHK(h, launch_reduce_partials(prb, wst));
        }
        if (overlap) HIPCHK(h, hipEventRecord(T.ev_wg[nbatch % TrainState::GSETS_MAX], wst));
        ++nbatch;
        pwb = WgradBatch{};
        prb = ReduceBatch{};
        pnw = punits = 0;
        plrows = 0;
        return MGN_OK;
    };
    // `fq` >= 0: first unit of the edge MLP of set fq with the factored first layer -- only the e block of W1 is unwound per edge
    // (gx[0] / gxadd[0] / xin[0] describe it); the v blocks follow per node from the summed rows of GZ1 (SGs, SGr) after this call.
    auto bwd_unit = [&](const TrainBlock& b, int64_t rows, int32_t ntiles, const float* g0, const float* g1, const int32_t* g1i, const size_t (&hb)[3],
                        float* const gx[3], const float* const gxadd[3], const float* const xin[3], const int32_t* const xi[3],
                        int fq = -1, const float* vin = nullptr, int lnslot = -1) -> int {
        const bool fact = fq >= 0;
        const bool wide = lnall && b.ln;
        const int64_t node_rows = fact ? N : 0;
        const int nin_k = fact ? 1 : b.nin;               // input blocks the kernel unwinds
        const int gs = overlap ? n_bwd % T.gsets : 0;
        // buffer set gs is re-used: the launch that took its last occupant's weight gradients must have run (launches on the second stream are
        // in order; their events are indexed by launch number)
        if (overlap && set_batch[gs] >= 0) {
            if (set_batch[gs] == nbatch && punits > 0)
                if (int rc = flush()) return rc;
            HIPCHK(h, hipStreamWaitEvent(st, T.ev_wg[set_batch[gs] % TrainState::GSETS_MAX], 0));
        }
        ++n_bwd;
        TrainBwdArgs a{};
        a.rows = rows; a.ntiles = ntiles;
        a.G0 = g0; a.G1 = g1; a.g1idx = g1i;
        a.H1 = A + hb[0]; a.H2 = A + hb[1]; a.Y = A + hb[2];
        a.W3T = Wt + b.W3T; a.W2T = Wt + b.W2T;
        for (int j = 0; j < nin_k; ++j) {
            const int jw = fact ? 2 : j;                   // factored: block 2 (e) of W1
            a.W1T[j] = (b.has_w1t && gx[j]) ? Wt + b.W1T[jw] : nullptr;
            a.GX[j] = gx[j];
            a.GXadd[j] = gxadd[j];
        }
        a.tabs = Wt + b.tabs;
        a.ln = b.ln ? 1 : 0;
        a.GT = A + T.GT[gs]; a.GXH = A + T.GXH[gs]; a.GY = A + T.GY[gs]; a.GZ2 = A + T.GZ2[gs]; a.GZ1 = A + T.GZ1[gs];
        // One stream (large meshes): the LayerNorm-parameter sums come from a job that re-reads G0 (+ G1) and Y -- they are intact until this
        // unit's weight-gradient launch has run -- and the row statistics, instead of from GT / G xhat rows written here and read there.
        const bool lnsum = b.ln && !wide && !overlap && rows > 0 && train_bwd_ln_sums(L, ntiles);     // the sums inside the backward kernel (per block) ...
        const bool lnjob = !lnsum && b.ln && !wide && !overlap && rows > 0 && wgrad_ln_jobs(L) && !train_uses_coop(L, ntiles);   // ... or a job that re-reads G0 / Y (the cooperative backward kernels write GT / G xhat)
        if (lnjob) { a.GT = nullptr; a.GXH = nullptr; a.LNROW = A + T.lnrow; }
        if (lnsum) { a.GT = nullptr; a.GXH = nullptr; a.LNSUM = A + T.lnsum; }
        if (b.ln && !wide && !lnsum && !T.need_gt && rows > 0) return fail(h, MGN_E_STATE, "training arena laid out without GT / GXH rows but a launch unit needs them");
        if (wide && rows > 0) {   // pullback of the whole-array LayerNorm: dgamma, dbeta and the two means first (two column reductions)
            HIPCHK(h, launch_lnall_bwd(g0, g1, g1i, A + hb[2], A + T.lnstats + (size_t)64 * lnslot, Wt + b.tabs + (size_t)T_GAMMA * L, rows, L,
                                       reinterpret_cast<double*>(A + T.lnpart), A + T.lnm, G + b.ggamma, G + b.gbeta, st));
            a.ln = 2;                   // the kernel maps G to the gradient at Y as it loads it
            a.LNS = A + T.lnstats + (size_t)64 * lnslot;
            a.LNM = A + T.lnm;
        }
        const bool sgr = fact && rows > 0 && train_bwd_fused_sgr(L, ntiles);   // SGr inside the launch (segmented scan over the receiver runs of GZ1)
        if (sgr) { a.SEG_RCV = sx[fq].rcv; a.SEG_OUT = A + T.SGr; a.SEG_CARRY = A + T.segcarry; }
        HIPCHK(h, launch_mlp_bwd(L, nin_k, a, st));
        if (lnsum)    // [blocks][2 L] -> [LNSUM_GROUPS][2 L], in order; the unit's reduction launch adds the groups
            HIPCHK(h, launch_colsum_groups(A + T.lnsum, (ntiles + 7) / 8, 2 * L, LNSUM_GROUPS, A + T.lnsum2, st));
        if (fact) {   // gather <-> segmented-sum duality on GZ1 itself: SGr[n] = sum of GZ1 over edges received by n, SGs: sent by n
            if (sgr) {
                HIPCHK(h, launch_seg_fixup(L, sx[fq].rowptr, A + T.segcarry, A + T.SGr, (int32_t)node_rows, st));
                HIPCHK(h, launch_segment_sum(L, A + T.GZ1[gs], sx[fq].rowptr_s, sx[fq].perm_s, nullptr, A + T.SGs, (int32_t)node_rows, st));
            } else
            HIPCHK(h, launch_segment_sum_pair(L, A + T.GZ1[gs], sx[fq].rowptr, sx[fq].rowptr_s, sx[fq].perm_s, A + T.SGr, A + T.SGs, (int32_t)node_rows, st));
        }
        // every parameter gradient of this unit: jobs of one batched weight-gradient launch + one batched (ordered) reduction
        const int64_t lrows_u = rows > node_rows ? rows : node_rows;   // a launch covers its longest job (node jobs of a factored edge MLP)
        const int64_t lrows = overlap ? lrows_all : lrows_u;           // (a group's launch: the longest job of the model; 128-row blocks either way)
        const int nb = wgrad_blocks(lrows);
        if (pwb.njobs + 10 > WGRAD_MAX_JOBS || prb.njobs + 14 > REDUCE_MAX_JOBS)   // (a unit adds at most 8 + 12 jobs)
            if (int rc = flush()) return rc;
        ++punits;
        set_batch[gs] = nbatch;
        if (nb == 0 || lrows_u == 0) return punits >= group ? flush() : MGN_OK;
        plrows = std::max(plrows, lrows);
        WgradBatch& wb = pwb;
        ReduceBatch& rb = prb;
        int& nw = pnw;
        const bool defer = T.defer_reduce && unit_no < T.defer_units;
        const int this_unit = unit_no++;
        int unw = 0, unb = 0;                                  // (deferred) weight / column-sum partial slots of this unit
        auto job = [&](const float* X, const int32_t* xi_, const float* Gm, long woff, int nrows, int cols, long boff, int bcols,
                       int64_t jrows = -1) {
            if (woff < 0 && boff < 0) return;              // identity slot
            WgradJob& j = wb.job[wb.njobs];
            if (jrows < 0) jrows = rows;
            const int nbj = wgrad_blocks_of_job(lrows, jrows);     // blocks of this launch that hold rows of the job
            j.X = X; j.xidx = xi_; j.G = Gm; j.rows = jrows;
            if (defer) {
                j.pw = woff >= 0 ? A + T.pw_all + ((size_t)this_unit * 5 + unw) * nb * L * L : nullptr;
                j.pb = boff >= 0 ? A + T.pb_all + ((size_t)this_unit * DEFER_PB_JOBS + unb) * nb * L : nullptr;
                if (woff >= 0) { deferred.push_back(ReduceJob{j.pw, nbj, (int64_t)L * L, nrows, cols, L, G + woff}); ++unw; }
                if (boff >= 0) { deferred.push_back(ReduceJob{j.pb, nbj, (int64_t)L, 1, bcols, L, G + boff}); ++unb; }
                ++wb.njobs;
                return;
            }
            j.pw = woff >= 0 ? A + T.pw + (size_t)nw * nb * L * L : nullptr;
            j.pb = boff >= 0 ? A + T.pb + (size_t)wb.njobs * nb * L : nullptr;
            if (woff >= 0) {
                rb.job[rb.njobs++] = ReduceJob{j.pw, nbj, (int64_t)L * L, nrows, cols, L, G + woff};
                ++nw;
            }
            if (boff >= 0) rb.job[rb.njobs++] = ReduceJob{j.pb, nbj, (int64_t)L, 1, bcols, L, G + boff};
            ++wb.njobs;
        };
        job(A + hb[1], nullptr, A + T.GY[gs], b.gW[2], L, b.out_cols, b.gb[2], b.out_cols);
        job(A + hb[0], nullptr, A + T.GZ2[gs], b.gW[1], L, L, b.gb[1], L);
        if (!fact) {
            for (int j = 0; j < b.nin; ++j)
                job(xin[j], xi[j], A + T.GZ1[gs], b.gW[0] + (long)j * L * L, b.in_rows, L, j == 0 ? b.gb[0] : -1, L);
        } else {   // dW1e = e^T GZ1 (+ db1) over the edges; dW1s = v^T SGs, dW1r = v^T SGr over the nodes
            job(xin[0], xi[0], A + T.GZ1[gs], b.gW[0] + (long)2 * L * L, L, L, b.gb[0], L);
            job(vin, nullptr, A + T.SGs, b.gW[0], L, L, -1, L, node_rows);
            job(vin, nullptr, A + T.SGr, b.gW[0] + (long)L * L, L, L, -1, L, node_rows);
        }
        if (lnsum) {
            const int ng = std::min(LNSUM_GROUPS, (int)((ntiles + 7) / 8));
            rb.job[rb.njobs++] = ReduceJob{A + T.lnsum2, ng, (int64_t)2 * L, 1, L, L, G + b.gbeta};
            rb.job[rb.njobs++] = ReduceJob{A + T.lnsum2 + L, ng, (int64_t)2 * L, 1, L, L, G + b.ggamma};
        } else if (lnjob) {
            WgradJob& j = wb.job[wb.njobs];
            const int nbj = wgrad_blocks_of_job(lrows, rows);
            j = WgradJob{};
            j.G = g0; j.rows = rows;
            j.Y = A + hb[2]; j.LNROW = A + T.lnrow; j.G1 = g1; j.g1idx = g1i;
            j.pb = A + T.pb + (size_t)wb.njobs * nb * L;
            j.pb2 = A + T.pb + (size_t)WGRAD_MAX_JOBS * nb * L;
            rb.job[rb.njobs++] = ReduceJob{j.pb, nbj, (int64_t)L, 1, L, L, G + b.gbeta};
            rb.job[rb.njobs++] = ReduceJob{j.pb2, nbj, (int64_t)L, 1, L, L, G + b.ggamma};
            ++wb.njobs;
        } else if (b.ln && !wide) {
            job(nullptr, nullptr, A + T.GXH[gs], -1, 0, 0, b.ggamma, L);
            job(nullptr, nullptr, A + T.GT[gs], -1, 0, 0, b.gbeta, L);
        }
        return punits >= group ? flush() : MGN_OK;
    };
    // one MLP: its second unit (if any) first, handing the gradient w.r.t. its input to the first through GXB
    auto bwd = [&](const TrainMlp& m, int64_t rows, int32_t ntiles, const float* g0, const float* g1, const int32_t* g1i, const Acts& act,
                   float* const gx[3], const float* const gxadd[3], const float* const xin[3], const int32_t* const xi[3],
                   int fq = -1, const float* vin = nullptr) -> int {
        if (m.nblk == 2) {
            float* gx1[3] = {A + T.GXB, nullptr, nullptr};
            const float* none[3] = {nullptr, nullptr, nullptr};
            const float* xin1[3] = {A + act.h[0][2], nullptr, nullptr};
            const int32_t* xi1[3] = {nullptr, nullptr, nullptr};
            if (int rc = bwd_unit(m.b[1], rows, ntiles, g0, g1, g1i, act.h[1], gx1, none, xin1, xi1, -1, nullptr, m.lnslot)) return rc;
            return bwd_unit(m.b[0], rows, ntiles, A + T.GXB, nullptr, nullptr, act.h[0], gx, gxadd, xin, xi, fq, vin, m.lnslot);
        }
        return bwd_unit(m.b[0], rows, ntiles, g0, g1, g1i, act.h[0], gx, gxadd, xin, xi, fq, vin, m.lnslot);
    };

    auto backward_launches = [&]() -> int {
    HIPCHK(h, hipMemsetAsync(G, 0, h->params.size() * 4, st));   // (inside the replayed sequence: G is the engine's own buffer)
    n_bwd = 0;
    unit_no = 0;
    deferred.clear();
    pwb = WgradBatch{}; prb = ReduceBatch{};
    pnw = punits = nbatch = 0;
    plrows = 0;
    for (int& v : set_batch) v = -1;
    int cur = 0;   // gV[cur], gE[q][ecur] hold the gradients w.r.t. the latents entering the part of the model already unwound
    {
        float* gx[3] = {A + T.gV[cur], nullptr, nullptr};
        const float* gxadd[3] = {nullptr, nullptr, nullptr};
        const float* xin[3] = {A + T.Vk[mps], nullptr, nullptr};
        const int32_t* xi[3] = {nullptr, nullptr, nullptr};
        if (int rc = bwd(T.m_de, N, nt_n, A + T.Gout, nullptr, nullptr, T.a_de, gx, gxadd, xin, xi)) return rc;
    }
    int ecur = 0;
    for (int q = 0; q < S; ++q) HIPCHK(h, hipMemsetAsync(A + T.gE[q][ecur], 0, (size_t)(sx[q].E > 0 ? sx[q].E : 1) * L * 4, st));
    for (int k = mps - 1; k >= 0; --k) {
        const int nxt = cur ^ 1, enxt = ecur ^ 1;
        if (!T.kept(k, mps)) {   // regenerate H1, H2, Y of the MLPs of this step from their kept inputs
            HIPCHK(h, fwd_node(k, nullptr, nullptr));
            for (int q = 0; q < S; ++q) HIPCHK(h, fwd_edge(q, k, nullptr, nullptr, nullptr));
        }
        {   // node MLP: v_{k+1} = v_k + MLP_v([v_k; agg_k (per set)])
            float* gx[3] = {A + T.gV[nxt], A + T.gAgg[0], S > 1 ? A + T.gAgg[1] : nullptr};
            const float* gxadd[3] = {A + T.gV[cur], nullptr, nullptr};
            const float* xin[3] = {A + T.Vk[k], A + T.agg[0][k], S > 1 ? A + T.agg[1][k] : nullptr};
            const int32_t* xi[3] = {nullptr, nullptr, nullptr};
            if (int rc = bwd(T.m_pn[k], N, nt_n, A + T.gV[cur], nullptr, nullptr, T.a_pn[k], gx, gxadd, xin, xi)) return rc;
        }
        for (int q = 0; q < S; ++q) {
            const TrainMlp& me = T.m_pe[q][k];
            const int64_t E = sx[q].E;
            if (!T.factored[q]) {   // edge MLP: e' feeds e_{k+1} = e_k + e' and agg_k[receiver]
                float* gx[3] = {A + T.GXs, A + T.GXr, A + T.gE[q][enxt]};
                const float* gxadd[3] = {nullptr, nullptr, A + T.gE[q][ecur]};
                const float* xin[3] = {A + T.Vk[k], A + T.Vk[k], A + T.Ek[q][k]};
                const int32_t* xi[3] = {sx[q].snd, sx[q].rcv, nullptr};
                if (int rc = bwd(me, E, sx[q].nt, A + T.gE[q][ecur], A + T.gAgg[q], sx[q].rcv, T.a_pe[q][k], gx, gxadd, xin, xi)) return rc;
                // gather duality: the gradients of v[receivers] / v[senders] are segmented sums over the receiver / sender CSR
                HIPCHK(h, launch_segment_sum2(L, A + T.GXr, sx[q].rowptr, A + T.GXs, sx[q].rowptr_s, sx[q].perm_s, A + T.gV[nxt], A + T.gV[nxt],
                                              (int32_t)N, st));
            } else {             // factored first layer: per edge only the e block; the v blocks per node from SGs / SGr
                float* gx[3] = {A + T.gE[q][enxt], nullptr, nullptr};
                const float* gxadd[3] = {A + T.gE[q][ecur], nullptr, nullptr};
                const float* xin[3] = {A + T.Ek[q][k], nullptr, nullptr};
                const int32_t* xi[3] = {nullptr, nullptr, nullptr};
                if (int rc = bwd(me, E, sx[q].nt, A + T.gE[q][ecur], A + T.gAgg[q], sx[q].rcv, T.a_pe[q][k], gx, gxadd, xin, xi, q, A + T.Vk[k]))
                    return rc;
                Lin2Args l2{};    // gV += SGs W1s^T + SGr W1r^T
                l2.rows = N; l2.ntiles = nt_n;
                l2.X0 = A + T.SGs; l2.X1 = A + T.SGr; l2.W0 = Wt + me.b[0].W1T[0]; l2.W1 = Wt + me.b[0].W1T[1];
                l2.ADD = A + T.gV[nxt]; l2.OUT0 = A + T.gV[nxt];
                HIPCHK(h, launch_lin2(L, l2, st));
            }
            if (E == 0) HIPCHK(h, hipMemsetAsync(A + T.gE[q][enxt], 0, (size_t)L * 4, st));
        }
        cur = nxt;
        ecur = enxt;
    }
    {
        float* gx_n[3] = {J.vjp ? A + T.gNF : nullptr, nullptr, nullptr};
        float* gx[3] = {nullptr, nullptr, nullptr};
        const float* gxadd[3] = {nullptr, nullptr, nullptr};
        const float* xin[3] = {A + T.nf_pad, nullptr, nullptr};
        const int32_t* xi[3] = {nullptr, nullptr, nullptr};
        if (int rc = bwd(T.m_en, N, nt_n, A + T.gV[cur], nullptr, nullptr, T.a_en, gx_n, gxadd, xin, xi)) return rc;
        for (int q = 0; q < S; ++q) {
            const float* xin_e[3] = {A + T.ef_pad[q], nullptr, nullptr};
            const int32_t* xi_e[3] = {sx[q].egid, nullptr, nullptr};
            if (int rc = bwd(T.m_ee[q], sx[q].E, sx[q].nt, A + T.gE[q][ecur], nullptr, nullptr, T.a_ee[q], gx, gxadd, xin_e, xi_e)) return rc;
        }
    }

    if (int rc = flush()) return rc;                     // the units left over from the last full group
    if (!deferred.empty()) {                             // every unit's reductions, REDUCE_MAX_JOBS per launch, behind the last weight-gradient launch
        static const int whatif = [] { const char* e = getenv("MGN_TRAIN_WHATIF"); return e ? atoi(e) : 0; }();
        hipStream_t wst = overlap ? T.aux : st;
        for (size_t i = 0; i < deferred.size() && !(whatif & 2); i += REDUCE_MAX_JOBS) {
            ReduceBatch rb{};
            for (size_t j = i; j < deferred.size() && j < i + REDUCE_MAX_JOBS; ++j) rb.job[rb.njobs++] = deferred[j];
            HIPCHK(h, launch_reduce_partials(rb, wst));
        }
        if (overlap) {
            HIPCHK(h, hipEventRecord(T.ev_wg[nbatch % TrainState::GSETS_MAX], wst));
            ++nbatch;
        }
    }
    if (overlap && nbatch > 0)                           // join: the second stream is in order, its last event covers all of it
        HIPCHK(h, hipStreamWaitEvent(st, T.ev_wg[(nbatch - 1) % TrainState::GSETS_MAX], 0));
    return MGN_OK;
    };
    if (int rc = graphed(J.vjp ? 2 : 1, backward_launches)) return rc;

    // ---- results
    std::vector<double> lp((size_t)nlb);
    HIPCHK(h, hipMemcpyAsync(J.grads, G, h->params.size() * 4, hipMemcpyDefault, st));
    if (!J.vjp) {
        HIPCHK(h, hipMemcpyAsync(lp.data(), T.loss.p, lp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    } else if (J.fvjp) {
        HIPCHK(h, launch_extract_cols(A + T.gNF, L, c.Fn, nullptr, A + T.io, N, st));
        HIPCHK(h, to_global(A + T.io, c.Fn));
        HIPCHK(h, hipMemcpyAsync(J.nfbar, A + T.io, (size_t)N * c.Fn * 4, hipMemcpyDefault, st));
        if (J.out) {
            HIPCHK(h, to_global(T.target.as<float>(), O));
            HIPCHK(h, hipMemcpyAsync(J.out, T.target.p, (size_t)N * O * 4, hipMemcpyDefault, st));
        }
    } else {
        // x enters through the node normaliser: xbar = (d / d nf)[:, 0:O] .* node_scale[0:O]
        HIPCHK(h, launch_extract_cols(A + T.gNF, L, O, h->have_nnorm ? nrm : nullptr, A + T.io, N, st));
        HIPCHK(h, to_global(A + T.io, O));
        HIPCHK(h, hipMemcpyAsync(J.xbar, A + T.io, (size_t)N * O * 4, hipMemcpyDefault, st));
        if (J.dxdt) {
            HIPCHK(h, to_global(T.target.as<float>(), O));
            HIPCHK(h, hipMemcpyAsync(J.dxdt, T.target.p, (size_t)N * O * 4, hipMemcpyDefault, st));
        }
    }
    HIPCHK(h, hipStreamSynchronize(st));
    if (!J.vjp) {
        double s = 0.0;
        for (double v : lp) s += v;
        *J.loss = (float)(s / (double)J.nmask);
    }
    return MGN_OK;
}


}  // namespace

extern "C" int mgn_step(mgn_handle* h, const float* nf, const float* ef, const float* target, const int32_t* mask, int64_t nmask,
                        int32_t mask_index_base, float* grads, size_t n_grads, float* loss) try {
    if (!h) return MGN_E_ARG;
    if (!nf || !target || !mask || !grads || !loss) return fail(h, MGN_E_ARG, "mgn_step: null argument");
    if (int rc = train_prepare(h, "mgn_step", n_grads)) return rc;
    if (!ef && h->g.set[0].E > 0) return fail(h, MGN_E_ARG, "mgn_step: null argument");
    if (nmask < 1) return fail(h, MGN_E_ARG, "mgn_step: empty mask");
    if (mask_index_base != 0 && mask_index_base != 1) return fail(h, MGN_E_ARG, "mgn_step: mask_index_base must be 0 or 1");
    for (int64_t i = 0; i < nmask; ++i) {
        const int64_t n = (int64_t)mask[i] - mask_index_base;
        if (n < 0 || n >= h->g.n_own) return fail(h, MGN_E_ARG, "mgn_step: mask entry %lld out of range", (long long)i);
    }
    TrainJob J;
    J.nf = nf; J.ef = ef; J.target = target; J.mask = mask; J.nmask = nmask; J.mask_index_base = mask_index_base;
    J.loss = loss; J.grads = grads;
    return train_run(h, J);
} MGN_CATCH(h)

extern "C" int mgn_ode_vjp(mgn_handle* h, const float* x, const float* node_type_onehot, const float* ef_raw, const float* val_mask,
                           const float* lambda, float* dxdt, float* xbar, float* grads, size_t n_grads) try {
    if (!h) return MGN_E_ARG;
    if (!x || !lambda || !xbar || !grads) return fail(h, MGN_E_ARG, "mgn_ode_vjp: null argument");
    if (int rc = train_prepare(h, "mgn_ode_vjp", n_grads)) return rc;
    const mgn_config& c = h->cfg;
    if (c.Fn < c.O) return fail(h, MGN_E_ARG, "mgn_ode_vjp: Fn < O");
    if ((c.Fn > c.O && !node_type_onehot) || (!ef_raw && h->g.set[0].E > 0)) return fail(h, MGN_E_ARG, "mgn_ode_vjp: null argument");
    TrainJob J;
    J.vjp = true;
    J.x = x; J.onehot = node_type_onehot; J.ef = ef_raw; J.val_mask = val_mask; J.lambda = lambda;
    J.dxdt = dxdt; J.xbar = xbar; J.grads = grads;
    return train_run(h, J);
} MGN_CATCH(h)

// =====================================================================================================================================
// mgn_config.ln_dims = MGN_LN_ALL: LayerNorm statistics over the whole (rows x L) output of an MLP -- what Lux 0.5's LayerNorm(shape)
// computes when GraphNetCore leaves it at dims = Colon() (reference Project.toml:15,40; julia/spec_probe.jl tells).  Every LayerNorm
// then couples all nodes / all edges, so nothing of it can be fused into a tile kernel: per MLP one launch of the training-forward
// kernel with its own LayerNorm off (train.hip: k_mlp_fwd, weights in training order), one grid-wide statistics pass (double, fixed
// order), one apply pass (+ residual).  Correct first: fp32-MFMA kernels, un-factored first edge layer, no kept activations.
// Serves mgn_forward (the model call, reference src/solve.jl:200) and mgn_processor_steps.
// =====================================================================================================================================
namespace {

struct LnAll {
    mgn_engine* h;
    TrainState& T;
    hipStream_t st;
    float* A;
    const float* Wt;
    int L;
    int64_t N, E;
    int32_t nt_n, nt_e;
    const int32_t *snd, *rcv, *rowptr, *egid;
    size_t V, Ecur, Y, Hb, agg, stats, part, nf_raw, nf_pad, ef_raw, ef_pad, tmp, E0, Pn, Qn;
    float eps_in, eps_out;
    int64_t slots = 0;           // (sum, sum of squares) slots the last MLP launch left in `part` (TrainFwdArgs::STATS)
    bool factored = false;       // large launches: the first edge layer per NODE (P = v W1s, Q = v W1r; launch_lin2) as in the training step

    // Y <- MLP(x) without LayerNorm / residual (launch units chained through Hb)
    hipError_t mlp(const TrainMlp& m, int64_t rows, int32_t ntiles, const float* x0, const int32_t* i0, const float* x1, const int32_t* i1,
                   const float* x2, float* yout) {
        for (int bi = 0; bi < m.nblk; ++bi) {
            const TrainBlock& b = m.b[bi];
            TrainFwdArgs a{};
            a.rows = rows; a.ntiles = ntiles;
            int nin = b.nin;
            if (bi == 0) {
                a.X[0] = x0; a.xidx[0] = i0; a.X[1] = x1; a.xidx[1] = i1; a.X[2] = x2;
                for (int j = 0; j < b.nin; ++j) a.W1[j] = Wt + b.W1[j];
            } else {
                a.X[0] = A + Hb;
                a.W1[0] = Wt + b.W1[0];
                nin = 1;
            }
            a.W2 = Wt + b.W2; a.W3 = Wt + b.W3; a.tabs = Wt + b.tabs;
            a.ln = 0;
            a.OUT = bi == m.nblk - 1 ? yout : A + Hb;
            if (bi == m.nblk - 1 && b.ln) { a.STATS = reinterpret_cast<double*>(A + part); slots = train_fwd_stat_slots(L, ntiles); }
            if (hipError_t e = launch_mlp_fwd(L, nin, a, st)) return e;
        }
        return hipSuccess;
    }
    // LayerNorm over all rows x L values of y with the MLP's gamma / beta: lnout = LN(y), out = resid + LN(y)
    hipError_t ln(const TrainMlp& m, const float* y, int64_t rows, const float* resid, float* out, float* lnout) {
        const TrainBlock& b = m.b[m.nblk - 1];
        const int64_t n = rows * L;
        if (n <= 0) return hipSuccess;
        if (hipError_t e = launch_array_stats_final(reinterpret_cast<const double*>(A + part), slots, n, eps_in, eps_out, A + stats, st)) return e;
        return launch_ln_all_apply(y, A + stats, Wt + b.tabs + (size_t)T_GAMMA * L, Wt + b.tabs + (size_t)T_BETA * L, resid, out, lnout, n, L, st);
    }
    // the edge half's LayerNorm, residual and aggregation in one pass over the receiver CSR (every edge has an owned receiver on one partition)
    hipError_t ln_edges(const TrainMlp& m, const float* y, float* e) {
        const TrainBlock& b = m.b[m.nblk - 1];
        if (hipError_t er = launch_array_stats_final(reinterpret_cast<const double*>(A + part), slots, E * L, eps_in, eps_out, A + stats, st)) return er;
        return launch_ln_all_apply_segsum(y, A + stats, Wt + b.tabs + (size_t)T_GAMMA * L, Wt + b.tabs + (size_t)T_BETA * L, e, rowptr, A + agg,
                                          (int32_t)N, L, st);
    }
    // one processor step on V / Ecur (engine order, row-major [rows][L])
    int step(int k) {
        float *v = A + V, *e = A + Ecur, *y = A + Y;
        if (E > 0 && factored) {
            const TrainMlp& m = T.m_pe[0][k];
            Lin2Args p{};
            p.rows = N; p.ntiles = nt_n;
            p.X0 = v; p.W0 = Wt + m.b[0].W1[0]; p.W1 = Wt + m.b[0].W1[1];
            p.OUT0 = A + Pn; p.OUT1 = A + Qn;
            HIPCHK(h, launch_lin2(L, p, st));
            for (int bi = 0; bi < m.nblk; ++bi) {
                const TrainBlock& b = m.b[bi];
                TrainFwdArgs a{};
                a.rows = E; a.ntiles = nt_e;
                if (bi == 0) {
                    a.X[0] = e; a.W1[0] = Wt + b.W1[2];
                    a.PRE[0] = A + Pn; a.preidx[0] = snd; a.PRE[1] = A + Qn; a.preidx[1] = rcv;
                } else {
                    a.X[0] = A + Hb; a.W1[0] = Wt + b.W1[0];
                }
                a.W2 = Wt + b.W2; a.W3 = Wt + b.W3; a.tabs = Wt + b.tabs;
                a.ln = 0;
                a.OUT = bi == m.nblk - 1 ? y : A + Hb;
                if (bi == m.nblk - 1) { a.STATS = reinterpret_cast<double*>(A + part); slots = train_fwd_stat_slots(L, nt_e); }
                HIPCHK(h, launch_mlp_fwd(L, 1, a, st));
            }
            HIPCHK(h, ln_edges(m, y, e));                                    // e <- e + LN(y), agg <- segmented sum of LN(y)
        } else if (E > 0) {
            HIPCHK(h, mlp(T.m_pe[0][k], E, nt_e, v, snd, v, rcv, e, y));
            HIPCHK(h, ln_edges(T.m_pe[0][k], y, e));
        } else {
            HIPCHK(h, launch_segment_sum(L, y, rowptr, nullptr, nullptr, A + agg, (int32_t)N, st));   // (no edges: zero aggregates)
        }
        HIPCHK(h, mlp(T.m_pn[k], N, nt_n, v, nullptr, A + agg, nullptr, nullptr, y));
        HIPCHK(h, ln(T.m_pn[k], y, N, v, v, nullptr));                       // v <- v + LN(MLP_v([v; agg]))
        return MGN_OK;
    }
};

int lnall_prepare(mgn_engine* h, const char* who, bool with_encoders) {
    if (!h) return MGN_E_ARG;
    if (h->host_only) return fail(h, MGN_E_HIP, "host-only handle (MGN_DEVICE_NONE): no compute path; create the handle on a HIP device");
    if (!h->have_params) return fail(h, MGN_E_STATE, "mgn_set_params has not been called");
    if (!h->have_graph) return fail(h, MGN_E_STATE, "mgn_set_graph has not been called");
    if (!h->train) h->train = new (std::nothrow) TrainState();
    if (!h->train) return fail(h, MGN_E_OOM, "host allocation failed");
    TrainState& T = *h->train;
    if (!T.packed)
        if (int rc = pack_training_weights(h)) return rc;
    (void)who; (void)with_encoders;
    return MGN_OK;
}

LnAll lnall_layout(mgn_engine* h, bool with_inputs, size_t& floats) {
    TrainState& T = *h->train;
    const LocalGraph& g = h->g;
    const mgn_config& c = h->cfg;
    LnAll X{h, T, h->stream, nullptr, T.w.as<float>(), c.L, g.n_own, g.set[0].e_local, 0, 0, nullptr, nullptr, nullptr, nullptr};
    X.nt_n = (int32_t)((X.N + TILE - 1) / TILE);
    X.nt_e = (int32_t)((X.E + TILE - 1) / TILE);
    const size_t NL = (size_t)(X.N > 0 ? X.N : 1) * c.L, EL = (size_t)(X.E > 0 ? X.E : 1) * c.L, ML = NL > EL ? NL : EL;
    size_t off = 0;
    auto take = [&](size_t n) { const size_t o = off; off += (n + 63) / 64 * 64; return o; };
    X.V = take(NL); X.Ecur = take(EL); X.Y = take(ML); X.Hb = T.nblk > 1 ? take(ML) : 0; X.agg = take(NL);
    X.factored = !train_uses_coop(128, X.nt_e) && X.E > 0;
    if (const char* e = getenv("MGN_TRAIN_FACTORED")) X.factored = atoi(e) != 0 && X.E > 0;
    X.Pn = X.Qn = 0;
    if (X.factored) { X.Pn = take(NL); X.Qn = take(NL); }
    X.stats = take(64);
    X.part = take(std::max<size_t>((size_t)4 * array_stats_blocks(), (size_t)16 * (size_t)std::max(X.nt_n, X.nt_e)));   // 2 doubles per slot
    X.tmp = take(ML);                                   // caller order <-> engine order staging
    X.nf_raw = X.nf_pad = X.ef_raw = X.ef_pad = X.E0 = 0;
    if (with_inputs) {
        X.nf_raw = take((size_t)X.N * c.Fn); X.nf_pad = take(NL);
        X.ef_raw = take((size_t)(X.E > 0 ? X.E : 1) * c.Fe); X.ef_pad = take(EL);
        X.E0 = take(EL);                                // the encoded edges of a trajectory (lnall_rhs_dev: the edge encoder runs once)
    }
    X.eps_in = c.ln_mode == MGN_LN_STD_EPS ? 0.f : 1e-5f;
    X.eps_out = c.ln_mode == MGN_LN_STD_EPS ? 1e-5f : 0.f;
    floats = off;
    return X;
}

int lnall_bind(mgn_engine* h, LnAll& X, size_t floats) {
    TrainState& T = *h->train;
    const LocalGraph& g = h->g;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, T.la.ensure(floats * 4));
    if (!T.la_ready) {
        std::vector<int32_t> eg((size_t)(X.E > 0 ? X.E : 1), 0);
        for (int64_t i = 0; i < X.E; ++i) eg[(size_t)i] = (int32_t)g.set[0].edge_gid[(size_t)i];
        HIPCHK(h, T.la_idx.ensure(eg.size() * 4));
        HIPCHK(h, hipMemcpy(T.la_idx.p, eg.data(), eg.size() * 4, hipMemcpyHostToDevice));
        T.la_ready = true;
    }
    X.A = T.la.as<float>();
    X.snd = h->es[0].d_snd.as<int32_t>();
    X.rcv = h->es[0].d_rcv.as<int32_t>();
    X.rowptr = h->es[0].d_rowptr.as<int32_t>();
    X.egid = T.la_idx.as<int32_t>();
    return MGN_OK;
}

}  // namespace

namespace mgn {

int lnall_forward(mgn_engine* h, const float* nf, const float* ef, float* out) {
    if (int rc = lnall_prepare(h, "mgn_forward", true)) return rc;
    h->lnall_edges = false;                             // (the arena is shared with the resident right-hand side)
    const mgn_config& c = h->cfg;
    const LocalGraph& g = h->g;
    if (!nf || !out || (!ef && g.set[0].E > 0)) return fail(h, MGN_E_ARG, "mgn_forward: null argument");
    size_t floats = 0;
    LnAll X = lnall_layout(h, true, floats);
    if (int rc = lnall_bind(h, X, floats)) return rc;
    TrainState& T = X.T;
    float* A = X.A;
    hipStream_t st = X.st;
    const int L = c.L;
    const int32_t* ngid = h->d_own_gid.as<int32_t>();
    // node features: caller's rows -> engine rows, padded to L; edge features stay in the caller's order and are gathered by edge id
    HIPCHK(h, hipMemcpyAsync(A + X.tmp, nf, (size_t)X.N * c.Fn * 4, hipMemcpyDefault, st));
    HIPCHK(h, launch_permute_rows(A + X.nf_raw, A + X.tmp, ngid, X.N, c.Fn, false, st));
    HIPCHK(h, launch_affine_pad(A + X.nf_raw, c.Fn, nullptr, 0, nullptr, nullptr, A + X.nf_pad, L, X.N, st));
    if (X.E > 0) {
        HIPCHK(h, hipMemcpyAsync(A + X.ef_raw, ef, (size_t)g.set[0].E * c.Fe * 4, hipMemcpyDefault, st));
        HIPCHK(h, launch_affine_pad(A + X.ef_raw, c.Fe, nullptr, 0, nullptr, nullptr, A + X.ef_pad, L, g.set[0].E, st));
    }
    // encoders
    HIPCHK(h, X.mlp(T.m_en, X.N, X.nt_n, A + X.nf_pad, nullptr, nullptr, nullptr, nullptr, A + X.Y));
    HIPCHK(h, X.ln(T.m_en, A + X.Y, X.N, nullptr, A + X.V, nullptr));
    if (X.E > 0) {
        HIPCHK(h, X.mlp(T.m_ee[0], X.E, X.nt_e, A + X.ef_pad, X.egid, nullptr, nullptr, nullptr, A + X.Y));
        HIPCHK(h, X.ln(T.m_ee[0], A + X.Y, X.E, nullptr, A + X.Ecur, nullptr));
    }
    for (int k = 0; k < c.mps; ++k)
        if (int rc = X.step(k)) return rc;
    // decoder (no LayerNorm): the first O columns of its output, back in the caller's row order
    HIPCHK(h, X.mlp(T.m_de, X.N, X.nt_n, A + X.V, nullptr, nullptr, nullptr, nullptr, A + X.Y));
    HIPCHK(h, launch_extract_cols(A + X.Y, L, c.O, nullptr, A + X.agg, X.N, st));
    HIPCHK(h, launch_permute_rows(A + X.tmp, A + X.agg, ngid, X.N, c.O, true, st));
    HIPCHK(h, hipMemcpyAsync(out, A + X.tmp, (size_t)X.N * c.O * 4, hipMemcpyDefault, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return MGN_OK;
}

int lnall_processor_steps(mgn_engine* h, float* v, float* e, int32_t nsteps) {
    if (int rc = lnall_prepare(h, "mgn_processor_steps", false)) return rc;
    h->lnall_edges = false;
    const mgn_config& c = h->cfg;
    const LocalGraph& g = h->g;
    if (!v || (!e && g.set[0].E > 0)) return fail(h, MGN_E_ARG, "mgn_processor_steps: null argument");
    if (nsteps < 0 || nsteps > c.mps) return fail(h, MGN_E_ARG, "mgn_processor_steps: nsteps out of range");
    size_t floats = 0;
    LnAll X = lnall_layout(h, false, floats);
    if (int rc = lnall_bind(h, X, floats)) return rc;
    float* A = X.A;
    hipStream_t st = X.st;
    const int L = c.L;
    const int32_t* ngid = h->d_own_gid.as<int32_t>();
    HIPCHK(h, hipMemcpyAsync(A + X.tmp, v, (size_t)X.N * L * 4, hipMemcpyDefault, st));
    HIPCHK(h, launch_permute_rows(A + X.V, A + X.tmp, ngid, X.N, L, false, st));
    if (X.E > 0) {
        HIPCHK(h, hipMemcpyAsync(A + X.tmp, e, (size_t)X.E * L * 4, hipMemcpyDefault, st));
        HIPCHK(h, launch_permute_rows(A + X.Ecur, A + X.tmp, X.egid, X.E, L, false, st));
    }
    for (int k = 0; k < nsteps; ++k)
        if (int rc = X.step(k)) return rc;
    HIPCHK(h, launch_permute_rows(A + X.tmp, A + X.V, ngid, X.N, L, true, st));
    HIPCHK(h, hipMemcpyAsync(v, A + X.tmp, (size_t)X.N * L * 4, hipMemcpyDefault, st));
    if (X.E > 0) {
        HIPCHK(h, hipStreamSynchronize(st));
        HIPCHK(h, launch_permute_rows(A + X.tmp, A + X.Ecur, X.egid, X.E, L, true, st));
        HIPCHK(h, hipMemcpyAsync(e, A + X.tmp, (size_t)X.E * L * 4, hipMemcpyDefault, st));
    }
    HIPCHK(h, hipStreamSynchronize(st));
    return MGN_OK;
}

// The right-hand side on resident inputs under ln_dims = MGN_LN_ALL -- what encode_impl + run_processor + decode_impl (mgn_api.cpp) are to
// the fused kernels: node inputs [state (srcA, in_wa columns) | static (d_nfB, in_wb columns)] and raw edge features as upload_inputs left
// them (caller's order with own_gid / edge_gid, or the engine's order), build_graph's normalisers, the model, inverse_data, val_mask;
// out [n_own][O] in the engine's order.  reuse_edges: the encoded edge latents of the trajectory are taken from the arena (static edge
// features, frozen e_norm: the edge encoder runs once per trajectory).  Launches only -- no host copy, no synchronisation -- once
// lnall_rhs_prepare has bound the arena, so the rollout driver can capture it.
int lnall_rhs_prepare(mgn_engine* h) {
    if (int rc = lnall_prepare(h, "right-hand side", true)) return rc;
    size_t floats = 0;
    LnAll X = lnall_layout(h, true, floats);
    return lnall_bind(h, X, floats);
}

int lnall_rhs_dev(mgn_engine* h, const float* srcA, float* out, bool reuse_edges) {
    const mgn_config& c = h->cfg;
    size_t floats = 0;
    LnAll X = lnall_layout(h, true, floats);
    TrainState& T = X.T;
    if (!T.la.p || T.la.bytes < floats * 4 || !T.la_ready) return fail(h, MGN_E_STATE, "whole-array LayerNorm: the right-hand side was not prepared");
    X.A = T.la.as<float>();
    X.snd = h->es[0].d_snd.as<int32_t>();
    X.rcv = h->es[0].d_rcv.as<int32_t>();
    X.rowptr = h->es[0].d_rowptr.as<int32_t>();
    X.egid = T.la_idx.as<int32_t>();
    float* A = X.A;
    hipStream_t st = X.st;
    const int L = c.L;
    const float* nrm = h->norms.as<float>();
    const int32_t* ngid = h->in_local ? nullptr : h->d_own_gid.as<int32_t>();
    const float* ns = h->have_nnorm ? nrm : nullptr;
    // node inputs: [srcA | srcB] normalised and padded to L, rows in the engine's order
    float* padded = ngid ? A + X.tmp : A + X.nf_pad;
    HIPCHK(h, launch_affine_pad(srcA, h->in_wa, h->d_nfB.as<float>(), h->in_wb, ns, ns ? ns + c.Fn : nullptr, padded, L, X.N, st));
    if (ngid) HIPCHK(h, launch_permute_rows(A + X.nf_pad, A + X.tmp, ngid, X.N, L, false, st));
    HIPCHK(h, X.mlp(T.m_en, X.N, X.nt_n, A + X.nf_pad, nullptr, nullptr, nullptr, nullptr, A + X.Y));
    HIPCHK(h, X.ln(T.m_en, A + X.Y, X.N, nullptr, A + X.V, nullptr));
    if (X.E > 0) {
        if (!reuse_edges) {
            const float* es = h->have_enorm ? nrm + 2 * c.Fn : nullptr;
            HIPCHK(h, launch_affine_pad(h->es[0].d_ef.as<float>(), c.Fe, nullptr, 0, es, es ? es + c.Fe : nullptr, A + X.ef_pad, L, X.E, st));
            HIPCHK(h, X.mlp(T.m_ee[0], X.E, X.nt_e, A + X.ef_pad, h->in_local ? nullptr : X.egid, nullptr, nullptr, nullptr, A + X.Y));
            HIPCHK(h, X.ln(T.m_ee[0], A + X.Y, X.E, nullptr, A + X.E0, nullptr));
        }
        HIPCHK(h, hipMemcpyAsync(A + X.Ecur, A + X.E0, (size_t)X.E * L * 4, hipMemcpyDeviceToDevice, st));
    }
    for (int k = 0; k < c.mps; ++k)
        if (int rc = X.step(k)) return rc;
    HIPCHK(h, X.mlp(T.m_de, X.N, X.nt_n, A + X.V, nullptr, nullptr, nullptr, nullptr, A + X.Y));
    const float* os = h->have_onorm ? nrm + 2 * c.Fn + 2 * c.Fe : nullptr;
    HIPCHK(h, launch_rhs_epilogue(A + X.Y, L, c.O, os, os ? os + c.O : nullptr, h->have_mask ? h->d_mask.as<float>() : nullptr,
                                  h->d_own_gid.as<int32_t>(), out, X.N, st));
    return MGN_OK;
}

}  // namespace mgn

// Pullback of mgn_forward == the model call `mgn.model(graph, ps, st)` at reference src/solve.jl:200 (what a ChainRulesCore.rrule of
// the Julia shim's model function returns to Zygote inside the pullback of ode_func_train, src/strategies.jl:183-195): given the
// cotangent ybar of the output, nfbar = ybar^T d out / d nf (all Fn columns) and grads = ybar^T d out / d ps.
extern "C" int mgn_forward_vjp(mgn_handle* h, const float* nf, const float* ef, const float* ybar, float* out, float* nfbar, float* grads,
                               size_t n_grads) try {
    if (!h) return MGN_E_ARG;
    if (!nf || !ybar || !nfbar || !grads) return fail(h, MGN_E_ARG, "mgn_forward_vjp: null argument");
    if (int rc = train_prepare(h, "mgn_forward_vjp", n_grads)) return rc;
    if (!ef && h->g.set[0].E > 0) return fail(h, MGN_E_ARG, "mgn_forward_vjp: null argument");
    TrainJob J;
    J.vjp = true;
    J.fvjp = true;
    J.nf = nf; J.ef = ef; J.lambda = ybar; J.out = out; J.nfbar = nfbar; J.grads = grads;
    return train_run(h, J);
} MGN_CATCH(h)

// Online-normaliser accumulation (GraphNetCore NormaliserOnline, used at reference src/MeshGraphNets.jl:92,193-199 and inside
// build_graph): per-feature sum and sum of squares of x [rows][dim], in double.
extern "C" int mgn_feature_stats(mgn_handle* h, const float* x, int64_t rows, int32_t dim, double* sum, double* sum_squares) try {
    if (!h) return MGN_E_ARG;
    if (int rc = need(h, false, false)) return rc;
    if (!x || !sum || !sum_squares || rows < 0 || dim < 1) return fail(h, MGN_E_ARG, "mgn_feature_stats: bad argument");
    for (int f = 0; f < dim; ++f) sum[f] = sum_squares[f] = 0.0;
    const int nb = stats_blocks(rows);
    if (nb == 0) return MGN_OK;
    const size_t xbytes = (size_t)rows * dim * 4, pbytes = (size_t)nb * 2 * dim * sizeof(double);
    HIPCHK(h, h->stage.ensure(xbytes + pbytes + 64));
    float* dx = h->stage.as<float>();
    double* dp = reinterpret_cast<double*>(reinterpret_cast<char*>(h->stage.p) + (xbytes + 63) / 64 * 64);
    HIPCHK(h, hipMemcpyAsync(dx, x, xbytes, hipMemcpyDefault, h->stream));       // host or device source
    HIPCHK(h, launch_col_stats(dx, rows, dim, dp, h->stream));
    std::vector<double> part((size_t)nb * 2 * dim);
    HIPCHK(h, hipMemcpyAsync(part.data(), dp, pbytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < nb; ++b)
        for (int f = 0; f < dim; ++f) {
            sum[f] += part[((size_t)b * 2 + 0) * dim + f];
            sum_squares[f] += part[((size_t)b * 2 + 1) * dim + f];
        }
    return MGN_OK;
} MGN_CATCH(h)

// tests / bench: how many processor steps of the training arena keep their activations (-1: no training arena yet)
extern "C" int mgn_debug_train_keep_steps(mgn_handle* h) { return (h && h->train && h->train->graph_ready) ? h->train->keep_steps : -1; }
