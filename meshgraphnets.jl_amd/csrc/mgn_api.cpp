// C ABI of the engine (include/mgn_hip.h).  Host orchestration only: parameter repacking into MFMA
// fragment order, device buffers, kernel sequencing.  All arithmetic is in kernels.hip; there is no
// CPU compute path here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "engine_internal.h"
#include "train.h"
#include "graph_dev.h"

using namespace mgn;


namespace {

thread_local std::string g_create_error;

}  // namespace

namespace mgn {

int fail(mgn_engine* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

}  // namespace mgn

namespace {

bool cfg_ok(const mgn_config* c, std::string& why) {
    if (!c) { why = "null config"; return false; }
    if (c->Fn < 1 || c->Fe < 1 || c->O < 1) { why = "Fn, Fe, O must be >= 1"; return false; }
    if (c->L != 32 && c->L != 64 && c->L != 128) { why = "L must be 32, 64 or 128 on the HIP path"; return false; }
    if (c->hidden_layers < 1 || c->hidden_layers > 4) { why = "hidden_layers must be 1 .. 4 on the HIP path"; return false; }
    if (c->mps < 1) { why = "mps must be >= 1"; return false; }
    if (c->dtype != MGN_F32 && c->dtype != MGN_BF16) { why = "dtype must be MGN_F32 or MGN_BF16"; return false; }
    if (c->dtype == MGN_BF16 && (c->L != 128 || c->hidden_layers != 2)) { why = "MGN_BF16 is implemented for L = 128, hidden_layers = 2"; return false; }
    if (c->nranks < 1 || c->rank < 0 || c->rank >= c->nranks) { why = "bad rank/nranks"; return false; }
    if (c->n_edge_sets < 0 || c->n_edge_sets > MAX_EDGE_SETS) { why = "n_edge_sets must be 0, 1 or 2"; return false; }
    if (c->n_edge_sets == 2 && c->Fe2 < 1) { why = "Fe2 must be >= 1 with two edge sets"; return false; }
    if (c->ln_mode != MGN_LN_VAR_EPS && c->ln_mode != MGN_LN_STD_EPS) { why = "ln_mode must be MGN_LN_VAR_EPS or MGN_LN_STD_EPS"; return false; }
    if (c->ln_dims != MGN_LN_ROWS && c->ln_dims != MGN_LN_ALL) { why = "ln_dims must be MGN_LN_ROWS or MGN_LN_ALL"; return false; }
    if (c->ln_dims == MGN_LN_ALL && (c->dtype != MGN_F32 || c->nranks != 1 || c->n_edge_sets > 1)) {
        why = "ln_dims = MGN_LN_ALL (whole-array LayerNorm) runs in fp32 on one partition with one edge set";
        return false;
    }
    return true;
}

// MGN-spec MLP: Dense(in -> L) . ReLU -> [Dense(L -> L) . ReLU] x (hidden_layers - 1) -> Dense(L -> out) [+ LayerNorm]
size_t mlp_layout(MlpOff& m, size_t off, int in, int L, int out, bool ln, int hidden_layers) {
    int dims[MAX_DENSE + 1];
    const int nl = hidden_layers + 1;
    dims[0] = in;
    for (int i = 1; i < nl; ++i) dims[i] = L;
    dims[nl] = out;
    m.in = in;
    m.out = out;
    m.ln = ln;
    m.nl = nl;
    for (int i = 0; i < nl; ++i) {
        m.W[i] = off;
        off += (size_t)dims[i] * dims[i + 1];
        m.b[i] = off;
        off += dims[i + 1];
    }
    if (ln) {
        m.gamma = off;
        off += out;
        m.beta = off;
        off += out;
    }
    return off;
}

size_t layout_all(mgn_engine* h) {
    const mgn_config& c = h->cfg;
    size_t off = 0;
    off = mlp_layout(h->enc_node, off, c.Fn, c.L, c.L, true, c.hidden_layers);
    // MGN-spec order: encoder-node, encoder-edge per set, (edge MLP per set, node MLP) x mps, decoder
    h->nsets = c.n_edge_sets == 2 ? 2 : 1;
    h->es[0].Fe = c.Fe;
    h->es[1].Fe = c.Fe2;
    for (int q = 0; q < h->nsets; ++q) {
        off = mlp_layout(h->es[q].enc, off, h->es[q].Fe, c.L, c.L, true, c.hidden_layers);
        h->es[q].pe.resize(c.mps);
    }
    h->pn.resize(c.mps);
    for (int k = 0; k < c.mps; ++k) {
        for (int q = 0; q < h->nsets; ++q) off = mlp_layout(h->es[q].pe[k], off, 3 * c.L, c.L, c.L, true, c.hidden_layers);
        off = mlp_layout(h->pn[k], off, (1 + h->nsets) * c.L, c.L, c.L, true, c.hidden_layers);
    }
    off = mlp_layout(h->dec, off, c.L, c.L, c.O, false, c.hidden_layers);
    return off;
}

inline int phi(int j, int hh) { return 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * hh; }

}  // namespace
namespace mgn {
// L x L chunk of W (row-major [K][ldw], rows kbase.., all L output columns) -> fragment order
void pack_chunk(float* dst, const float* W, int ldw, int kbase, int L) {
    const int NT = L / 32, J = L / 2;
    for (int j = 0; j < J; ++j)
        for (int lane = 0; lane < 64; ++lane) {
            const int hh = lane >> 5, i = lane & 31;
            const float* wrow = W + (size_t)(kbase + phi(j, hh)) * ldw;
            for (int t = 0; t < NT; ++t) dst[((size_t)j * 64 + lane) * NT + t] = wrow[32 * t + i];
        }
}
// t-major copy of a packed chunk, four consecutive k-steps per lane contiguous: [t][j/4][lane][j%4] (cooperative kernels)
void pack_chunk_tmajor(float* dst, const float* frag, int L) {
    const int NT = L / 32, J = L / 2;
    for (int t = 0; t < NT; ++t)
        for (int j = 0; j < J; ++j)
            for (int lane = 0; lane < 64; ++lane)
                dst[(((size_t)t * (J / 4) + j / 4) * 64 + lane) * 4 + (j & 3)] = frag[((size_t)j * 64 + lane) * NT + t];
}
// 128 x 128 chunk of W (row-major [K][ldw], rows kbase..) -> v_mfma_f32_16x16x4_f32 fragment order of the 16-row cooperative
// kernels: wave w owns output blocks 2w, 2w+1 (16 features each); k-step (bb, i) contracts the four input features
// 16 bb + 4 q + i (q = lane >> 4), i.e. register (bb, i) of every lane's row fragment:
//   dst[(((w*8 + bb)*2 + j)*64 + lane)*4 + i] = W[kbase + 16 bb + 4 (lane>>4) + i][16 (2w + j) + (lane & 15)]
void pack_chunk16(float* dst, const float* W, int ldw, int kbase) {
    for (int w = 0; w < 4; ++w)
        for (int bb = 0; bb < 8; ++bb)
            for (int j = 0; j < 2; ++j)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i)
                        dst[((((size_t)w * 8 + bb) * 2 + j) * 64 + lane) * 4 + i] =
                            W[(size_t)(kbase + 16 * bb + 4 * (lane >> 4) + i) * ldw + 16 * (2 * w + j) + (lane & 15)];
}
// vector of L values (stride between consecutive features = stride) -> table fragment order
void pack_tab(float* dst, const float* vec, int L, int stride) {
    for (int m = 0; m < L / 8; ++m)
        for (int hh = 0; hh < 2; ++hh)
            for (int i = 0; i < 4; ++i)
                dst[(m * 2 + hh) * 4 + i] = vec ? vec[(size_t)(32 * (m >> 2) + 8 * (m & 3) + 4 * hh + i) * stride] : 0.f;
}
}  // namespace mgn
namespace {

inline uint16_t f32_to_bf16(float f) {   // round to nearest even (inputs are finite weights / latents)
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// feature held by element j of piece (s, hh) of a bf16 row fragment (see kernels.hip, bf16 section)
inline int bf_feature(int sidx, int hh, int j) { return 32 * (sidx >> 1) + 16 * (sidx & 1) + 8 * (j >> 2) + 4 * hh + (j & 3); }
inline float bf16_to_f32(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// the same chunk as three bf16 pieces with w = hi + mid + lo exactly (dst: 3 x 16384, each in the fragment order below)
void pack_chunk_bf16(uint16_t* dst, const float* W, int ldw, int kbase);
void pack_chunk16_bf16(uint16_t* dst, const float* W, int ldw, int kbase);
void pack_chunk_split(uint16_t* dst, const float* W, int ldw, int kbase, bool frag16 = false) {
    std::vector<float> r1((size_t)128 * 128), r2((size_t)128 * 128), w0((size_t)128 * 128);
    for (int k = 0; k < 128; ++k)
        for (int n = 0; n < 128; ++n) {
            const float w = W[(size_t)(kbase + k) * ldw + n];
            const float a = bf16_to_f32(f32_to_bf16(w));
            const float ra = w - a;
            const float b = bf16_to_f32(f32_to_bf16(ra));
            w0[(size_t)k * 128 + n] = w;
            r1[(size_t)k * 128 + n] = ra;
            r2[(size_t)k * 128 + n] = ra - b;
        }
    auto pk = frag16 ? pack_chunk16_bf16 : pack_chunk_bf16;
    pk(dst, w0.data(), 128, 0);
    pk(dst + 16384, r1.data(), 128, 0);
    pk(dst + 2 * 16384, r2.data(), 128, 0);
}
// the same chunk times a power of two as two fp16 pieces, w scale = hi + lo (+ <= 2^-23 relative; split_common.hpp), both round to nearest
// even (dst: 2 x 16384, the bf16 pieces' fragment order)
void pack_chunk_split_h(uint16_t* dst, const float* W, int ldw, int kbase, float scale, bool frag16 = false) {
    auto put = [&](size_t at, int k, int n) {
        const float ws = W[(size_t)(kbase + k) * ldw + n] * scale;
        const _Float16 hi = (_Float16)ws;
        const _Float16 lo = (_Float16)(ws - (float)hi);
        memcpy(&dst[at], &hi, 2);
        memcpy(&dst[16384 + at], &lo, 2);
    };
    if (frag16) {                                        // [ks][ob][lane][8] (pack_chunk16_bf16's order)
        for (int ks = 0; ks < 4; ++ks)
            for (int ob = 0; ob < 8; ++ob)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j)
                        put((((size_t)ks * 8 + ob) * 64 + lane) * 8 + j, 16 * (2 * ks + (j >> 2)) + 4 * (lane >> 4) + (j & 3), 16 * ob + (lane & 15));
        return;
    }
    for (int sidx = 0; sidx < 8; ++sidx)
        for (int t = 0; t < 4; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    put((((size_t)sidx * 4 + t) * 64 + lane) * 8 + j, bf_feature(sidx, lane >> 5, j), 32 * t + (lane & 31));
}
// the A fragments of v_mfma_f32_16x16x32_bf16 (the 16-row kernels of kernels.hip on the split path): step (ks, ob) = output block ob of k-step ks; lane
// (r16 = lane & 15, g = lane >> 4) holds output 16 ob + r16, its element j input 16 (2 ks + (j >> 2)) + 4 g + (j & 3) -- the order in
// which a lane's two accumulator blocks 2 ks, 2 ks + 1 of the layer before hold them
void pack_chunk16_bf16(uint16_t* dst, const float* W, int ldw, int kbase) {
    for (int ks = 0; ks < 4; ++ks)
        for (int ob = 0; ob < 8; ++ob)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * (2 * ks + (j >> 2)) + 4 * (lane >> 4) + (j & 3), n = 16 * ob + (lane & 15);
                    dst[(((size_t)ks * 8 + ob) * 64 + lane) * 8 + j] = f32_to_bf16(W[(size_t)(kbase + k) * ldw + n]);
                }
}
// 128 x 128 chunk of W (row-major [K][ldw], rows kbase..) -> bf16 fragment order [s][t][lane][8]
void pack_chunk_bf16(uint16_t* dst, const float* W, int ldw, int kbase) {
    for (int sidx = 0; sidx < 8; ++sidx)
        for (int t = 0; t < 4; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int hh = lane >> 5, i = lane & 31;
                    dst[(((size_t)sidx * 4 + t) * 64 + lane) * 8 + j] = f32_to_bf16(W[(size_t)(kbase + bf_feature(sidx, hh, j)) * ldw + 32 * t + i]);
                }
}

}  // namespace
namespace mgn {
int need(mgn_engine* h, bool params, bool graph, bool packed, bool lnall_ok) {
    if (!h) return MGN_E_ARG;
    if (h->host_only) return fail(h, MGN_E_HIP, "host-only handle (MGN_DEVICE_NONE): no compute path; create the handle on a HIP device");
    // whole-array LayerNorm couples every row of an MLP's output: the fused kernels behind the other compute entry points cannot
    // compute it (mgn_forward and mgn_processor_steps branch off to the unfused driver before they get here)
    if (params && graph && h->cfg.ln_dims == MGN_LN_ALL && !lnall_ok)
        return fail(h, MGN_E_UNSUPPORTED, "ln_dims = MGN_LN_ALL (whole-array LayerNorm) is not served by this entry point (the staged mgn_fwd_* / mgn_proc_* calls run the fused kernels only)");
    if (params && !h->have_params) return fail(h, MGN_E_STATE, "mgn_set_params has not been called");
    if (graph && !h->have_graph) return fail(h, MGN_E_STATE, "mgn_set_graph has not been called");
    if (params && packed && !h->packed_ok) return pack_inference_weights(h);
    return MGN_OK;
}
}  // namespace mgn
namespace {

struct ProfScope {
    mgn_engine* h;
    int fam;
    hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t take(mgn_engine* h) {
        hipEvent_t e = nullptr;
        if (!h->event_pool.empty()) {
            e = h->event_pool.back();
            h->event_pool.pop_back();
        } else if (hipEventCreate(&e) != hipSuccess) {
            e = nullptr;
        }
        return e;
    }
    ProfScope(mgn_engine* h_, int fam_) : h(h_), fam(fam_) {
        if (!h->prof) return;
        a = take(h);
        b = take(h);
        if (a && b) (void)hipEventRecord(a, h->stream);
    }
    ~ProfScope() {
        if (a && b) {
            (void)hipEventRecord(b, h->stream);
            h->recs.push_back({fam, a, b});
        }
    }
};

const float* W(const mgn_engine* h, size_t off) { return h->wfrag.as<float>() + off; }

// tile-major storage: rows padded to whole 32-row tiles
size_t tile_floats(int64_t ntiles, int L) { return (size_t)ntiles * TILE * L; }


inline int64_t tiles_or_one(int32_t nt) { return nt > 0 ? nt : 1; }

// per-set buffers (sized by that set's local edges and by the owned + halo nodes)
int alloc_edge_set(mgn_engine* h, int q) {
    const int L = h->cfg.L;
    const LocalGraph& g = h->g;
    auto& es = h->es[q];
    const int64_t nte = tiles_or_one(es.ntiles_e), ntn = tiles_or_one(h->ntiles_n);
    struct { DevBuf* b; size_t bytes; } bufs[5] = {
        {&es.P, (size_t)(g.n_own + g.n_halo + 1 + 16) * L * 4}, {&es.Q, (size_t)(g.n_own + 1 + 16) * L * 4},    // (+ 16: rows live in blocks of eight)
        {&es.Elat, tile_floats(nte, L) * 4}, {&es.AGG, tile_floats(ntn, L) * 4}, {&es.CARRY, (size_t)(4 * nte + 1 + 16) * L * 4}};
    // (CARRY: two rows per 32-edge tile, or per 16-edge tile with the 16-row cooperative kernels; its last row stays zero)
    // padding rows of the tile-major arrays and the zero row of CARRY (its last row) must read as 0
    for (auto& b : bufs) {
        HIPCHK(h, b.b->ensure(b.bytes));
        HIPCHK(h, hipMemsetAsync(b.b->p, 0, b.bytes, h->stream));
    }
    if (h->cfg.dtype == MGN_BF16) {
        struct { DevBuf* b; size_t bytes; } bb[5] = {
            {&es.bElat, tile_floats(nte, L) * 2}, {&es.bAGG, tile_floats(ntn, L) * 2},
            {&es.bP, (size_t)(g.n_own + g.n_halo + 1 + 16) * L * 2}, {&es.bQ, (size_t)(g.n_own + 1 + 16) * L * 2},
            {&es.bCARRY, (size_t)(4 * nte + 1 + 16) * L * 2}};
        for (auto& b : bb) {
            HIPCHK(h, b.b->ensure(b.bytes));
            HIPCHK(h, hipMemsetAsync(b.b->p, 0, b.bytes, h->stream));
        }
    }
    return MGN_OK;
}

int alloc_latents(mgn_engine* h) {
    const int L = h->cfg.L;
    const LocalGraph& g = h->g;
    const int64_t ntn = tiles_or_one(h->ntiles_n);
    HIPCHK(h, h->V.ensure(tile_floats(ntn, L) * 4));
    HIPCHK(h, h->d_out.ensure((size_t)(g.n_own + 1) * h->cfg.O * 4));
    HIPCHK(h, h->d_sum.ensure(4 * sizeof(double)));
    HIPCHK(h, hipMemsetAsync(h->V.p, 0, tile_floats(ntn, L) * 4, h->stream));
    if (h->cfg.dtype == MGN_BF16) {
        HIPCHK(h, h->bV.ensure(tile_floats(ntn, L) * 2));
        HIPCHK(h, hipMemsetAsync(h->bV.p, 0, tile_floats(ntn, L) * 2, h->stream));
    }
    for (int q = 0; q < h->nsets; ++q)
        if (int rc = alloc_edge_set(h, q)) return rc;
    return MGN_OK;
}

void drop_graph(mgn_engine* h) {
    if (h->rhs_exec) (void)hipGraphExecDestroy(h->rhs_exec);
    h->rhs_exec = nullptr;
    h->rhs_warm = false;
    if (h->fwd_exec) (void)hipGraphExecDestroy(h->fwd_exec);
    h->fwd_exec = nullptr;
    h->fwd_warm = false;
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
    h->graph_nsteps = -1;
    h->graph_warm = -1;
}

// Small meshes: a launch sequence over buffers with fixed addresses runs eagerly once (per-kernel attributes are set outside
// of any capture), is captured on the next call and replayed afterwards.
template <typename F>
int run_graphed(mgn_engine* h, hipGraphExec_t& exec, bool& warm, F&& launches) {
    // (the legacy NULL stream -- mgn_set_stream(h, NULL) -- cannot be captured: eager there)
    const bool graphable = h->use_graph && !h->prof && h->stream != nullptr && launch_is_small(h->ntiles_n);
    if (graphable && exec) {
        HIPCHK(h, hipGraphLaunch(exec, h->stream));
        return MGN_OK;
    }
    if (!graphable || !warm) {
        warm = true;
        return launches();
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        h->use_graph = 0;               // a stream that cannot be captured: eager from here on
        return launches();
    }
    const int rc = launches();
    const hipError_t ce = hipStreamEndCapture(h->stream, &graph);
    if (rc != MGN_OK || ce != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr;
        h->use_graph = 0;               // eager from here on
        if (rc != MGN_OK) return rc;
        return launches();
    }
    (void)hipGraphDestroy(graph);
    HIPCHK(h, hipGraphLaunch(exec, h->stream));
    return MGN_OK;
}

// the resident right-hand side (mgn_set_static) and the hipGraph captured over its buffers go together
void invalidate_static(mgn_engine* h) {
    h->have_static = false;
    h->lnall_edges = false;
    if (h->rhs_exec) {
        if (!h->host_only) (void)hipStreamSynchronize(h->stream);
        (void)hipGraphExecDestroy(h->rhs_exec);
    }
    h->rhs_exec = nullptr;
    h->rhs_warm = false;
}

// 16-row cooperative tiles (v_mfma_f32_16x16x4_f32): both kernels of a processor step must agree (the carry rows are per 16-edge
// tile then), so the choice is made per handle and graph: fp32, L = 128, hidden_layers = 2, and the node launch and EVERY edge
// set's launch in the cooperative size range
int32_t use_c16(const mgn_engine* h) {
    // (bf16 mode included: the 16-row kernels then read and write the bf16 arrays and keep fp32 weights and arithmetic)
    if (!(coop16_enabled() && h->cfg.L == 128 && h->cfg.hidden_layers == 2 && get_kernel_path() != 4 && launch_is_small(h->ntiles_n)))
        return 0;
    const bool ring_hs = h->cfg.dtype == MGN_F32 && h->nsets == 1 && ring_hs_default();
    for (int q = 0; q < h->nsets; ++q)
        if (!(launch_is_small_edge(h->es[q].ntiles_e) && coop16_size(h->es[q].ntiles_e, h->ntiles_n, ring_hs))) return 0;
    return 1;
}

// GenMlp of an MLP: used when hidden_layers != 2 (or when tests force the GEN kernels, kernel path 4)
GenMlp gen_of(const mgn_engine* h, const GenOff& g, bool has_last) {
    GenMlp m{};
    const int nmid = h->cfg.hidden_layers - 1;
    for (int i = 0; i < nmid + (has_last ? 1 : 0); ++i) m.chunk[i] = W(h, g.ch[i]);
    m.tabs = W(h, g.tabs);
    m.nmid = nmid;
    m.use = (h->cfg.hidden_layers != 2 || get_kernel_path() == 4) ? 1 : 0;
    return m;
}

// step 0's edge launch can read its e rows from a second array (EdgeArgs::ElatSrc): one edge set on the 16-row kernels, one partition
static bool elat_src_ok(mgn_engine* h) {
    static const int on = [] { const char* e = getenv("MGN_RHS_ELAT_SRC"); return e ? atoi(e) : 1; }();   // 0: restore copy per right-hand side
    return on && h->nsets == 1 && h->cfg.nranks == 1 && h->cfg.L == 128 && h->cfg.hidden_layers == 2 && get_kernel_path() != 4 && use_c16(h);
}

EdgeArgs edge_args(mgn_engine* h, int k, int q = 0) {
    EdgeArgs a{};
    auto& es = h->es[q];
    a.snd = es.d_snd.as<int32_t>();
    a.rcv = es.d_rcv.as<int32_t>();
    a.E = h->g.set[q].e_local;
    a.ntiles = es.ntiles_e;
    a.bf = h->cfg.dtype == MGN_BF16 ? 1 : 0;            // (only the 16-row kernels are launched with fp32 EdgeArgs in bf16 mode)
    a.P = a.bf ? es.bP.as<float>() : es.P.as<float>();
    a.Q = a.bf ? es.bQ.as<float>() : es.Q.as<float>();
    a.Elat = a.bf ? es.bElat.as<float>() : es.Elat.as<float>();
    a.AGG = a.bf ? es.bAGG.as<float>() : es.AGG.as<float>();
    a.CARRY = a.bf ? es.bCARRY.as<float>() : es.CARRY.as<float>();
    for (int i = 0; i < 3; ++i) {
        a.chunk[i] = W(h, h->soff[k].e_ch[q][i]);
        a.chunk_t[i] = a.chunk[i] + (size_t)h->cfg.L * h->cfg.L;
    }
    a.tabs = W(h, h->soff[k].e_tabs[q]);
    a.gen = gen_of(h, h->soff[k].e_gen[q], true);
    const bool have_sp = k < (int)h->spoff.size() && h->wsp.p;
    for (int i = 0; i < 3; ++i) a.split[i] = (!a.bf && have_sp) ? h->wsp.as<uint16_t>() + h->spoff[k].e_ch[q][i] : nullptr;
    for (int i = 0; i < 3; ++i) a.split16[i] = have_sp ? h->wsp.as<uint16_t>() + h->spoff[k].e16_ch[q][i] : nullptr;
    const bool have_h = have_sp && h->spoff[k].have_h;
    for (int i = 0; i < 3; ++i) {
        a.splith[i] = (have_h && !a.bf) ? h->wsp.as<uint16_t>() + h->spoff[k].eh_ch[q][i] : nullptr;
        a.split16h[i] = have_h ? h->wsp.as<uint16_t>() + h->spoff[k].e16h_ch[q][i] : nullptr;
        a.h2_s[i] = have_h ? h->spoff[k].eh_s[q][i] : 1.f;
        a.h2_rs[i] = 1.f / a.h2_s[i];
    }
    a.h2_b2pos = have_h ? h->spoff[k].e_b2pos[q] : 0.f;
    a.c16 = use_c16(h);
    if (k == 0 && q == 0 && h->elat_src_override && a.c16 && !a.gen.use) a.ElatSrc = h->elat_src_override;
    a.stagger = h->stagger_edge;
    a.tile0 = 0;
    a.stamps = h->d_stamps.as<unsigned long long>();
    return a;
}

// tiles [0, tb) of set q hold every edge with a halo sender (they need the exchanged P rows); tiles [tb, ntiles) do not
inline int32_t boundary_tiles(const mgn_engine* h, int q) {
    const int64_t tb = (h->g.set[q].halo_span + TILE - 1) / TILE;
    return (int32_t)(tb < h->es[q].ntiles_e ? tb : h->es[q].ntiles_e);
}

// q: the edge set whose P,Q the projection part (modes 1, 2) writes
NodeArgs node_args(mgn_engine* h, int k, int mode, int q = 0) {
    NodeArgs a{};
    const size_t CH = (size_t)h->cfg.L * h->cfg.L;
    const auto& so = h->soff[k];
    a.n = h->g.n_own;
    a.ntiles = h->ntiles_n;
    a.rowptr = h->es[0].d_rowptr.as<int32_t>();
    const bool bf = h->cfg.dtype == MGN_BF16;           // (only the 16-row kernels are launched with fp32 NodeArgs in bf16 mode)
    a.bf = bf ? 1 : 0;
    a.V = bf ? h->bV.as<float>() : h->V.as<float>();
    a.AGG = bf ? h->es[0].bAGG.as<float>() : h->es[0].AGG.as<float>();
    a.CARRY = bf ? h->es[0].bCARRY.as<float>() : h->es[0].CARRY.as<float>();
    a.P = bf ? h->es[q].bP.as<float>() : h->es[q].P.as<float>();
    a.Q = bf ? h->es[q].bQ.as<float>() : h->es[q].Q.as<float>();
    for (int i = 0; i < 6; ++i) {
        a.chunk[i] = W(h, so.n_ch[i]);
        a.chunk_t[i] = a.chunk[i] + CH;
    }
    a.tabs = W(h, so.n_tabs);
    if (q == 1) {
        for (int i = 0; i < 2; ++i) {
            a.chunk[4 + i] = W(h, so.p1_ch[i]);
            a.chunk_t[4 + i] = a.chunk[4 + i] + CH;
        }
        a.tabs = W(h, so.p1_tabs);
    }
    if (h->nsets > 1 && mode != 2) {
        a.rowptr2 = h->es[1].d_rowptr.as<int32_t>();
        a.AGG2 = bf ? h->es[1].bAGG.as<float>() : h->es[1].AGG.as<float>();
        a.CARRY2 = bf ? h->es[1].bCARRY.as<float>() : h->es[1].CARRY.as<float>();
        a.zero_row2 = 4 * tiles_or_one(h->es[1].ntiles_e);
        a.chunk[6] = W(h, so.n_ch[6]);
        a.chunk_t[6] = a.chunk[6] + CH;
        if (q == 0) {                                   // the 16-row kernels project both sets in one launch (mode 1)
            a.P2 = bf ? h->es[1].bP.as<float>() : h->es[1].P.as<float>();
            a.Q2 = bf ? h->es[1].bQ.as<float>() : h->es[1].Q.as<float>();
            a.tabs2 = W(h, so.p1_tabs);
            for (int i = 0; i < 2; ++i) {
                a.chunk[7 + i] = W(h, so.p1_ch[i]);
                a.chunk_t[7 + i] = a.chunk[7 + i] + CH;
            }
        }
    }
    a.mode = mode;
    a.gen = gen_of(h, so.n_gen, true);
    a.stagger = h->stagger_node;
    a.zero_row = 4 * tiles_or_one(h->es[0].ntiles_e);
    a.c16 = use_c16(h);
    a.stamps = h->d_stamps.as<unsigned long long>();
    a.tile0 = 0;
    const bool sp16 = k < (int)h->spoff.size() && h->spoff[k].have_n && h->wsp.p;
    const bool sp = !bf && sp16;
    for (int i = 0; i < 6; ++i) a.split[i] = (sp && q == 0) ? h->wsp.as<uint16_t>() + h->spoff[k].n_ch[i] : nullptr;
    const bool sph = sp && q == 0 && h->nsets == 1 && h->spoff[k].have_h;        // two fp16 pieces, 32x32x16 order (fp32 storage, one edge set)
    const bool sph16 = sp16 && h->spoff[k].have_h;                               // ... 16x16x32 order (16-row kernels: both storage modes, both sets)
    for (int i = 0; i < 9; ++i) {
        a.split16h[i] = nullptr;
        a.h2_s[i] = 1.f;
    }
    for (int i = 0; i < 6; ++i) {
        a.splith[i] = sph ? h->wsp.as<uint16_t>() + h->spoff[k].nh_ch[i] : nullptr;
        if (sph16) {
            a.split16h[i] = h->wsp.as<uint16_t>() + h->spoff[k].n16h_ch[i];
            a.h2_s[i] = h->spoff[k].nh_s[i];
        }
    }
    if (sph16 && h->nsets == 2) {
        for (int i = 0; i < 3; ++i) {
            a.split16h[6 + i] = h->wsp.as<uint16_t>() + h->spoff[k].n16h_ch[6 + i];
            a.h2_s[6 + i] = h->spoff[k].nh_s[6 + i];
        }
        if (q == 1)                                     // the projection of set 1 (mode 2): its WP / WQ pieces in slots 4, 5
            for (int i = 0; i < 2; ++i) {
                a.split16h[4 + i] = a.split16h[7 + i];
                a.h2_s[4 + i] = a.h2_s[7 + i];
            }
    }
    for (int i = 0; i < 9; ++i) a.h2_rs[i] = 1.f / a.h2_s[i];
    a.h2_b2pos = sph ? h->spoff[k].n_b2pos : 0.f;
    if (sp && h->nsets == 2) {
        if (q == 1)                                     // the projection of set 1 (mode 2): its WP / WQ pieces
            for (int i = 0; i < 2; ++i) a.split[4 + i] = h->wsp.as<uint16_t>() + h->spoff[k].n2_ch[1 + i];
        a.split[6] = h->wsp.as<uint16_t>() + h->spoff[k].n2_ch[0];
    }
    // the same chunks in the 16x16x32 fragment order, in NodeArgs.chunk numbering (16-row cooperative kernels on the split path)
    for (int i = 0; i < 9; ++i) a.split16[i] = nullptr;
    if (sp16) {
        for (int i = 0; i < 6; ++i) a.split16[i] = h->wsp.as<uint16_t>() + h->spoff[k].n16_ch[i];
        if (h->nsets == 2) {
            for (int i = 0; i < 3; ++i) a.split16[6 + i] = h->wsp.as<uint16_t>() + h->spoff[k].n16_ch[6 + i];
            if (q == 1)
                for (int i = 0; i < 2; ++i) a.split16[4 + i] = a.split16[7 + i];
        }
    }
    return a;
}

const uint16_t* WB(const mgn_engine* h, size_t off) { return h->wbf.as<uint16_t>() + off; }

BfEdgeArgs bf_edge_args(mgn_engine* h, int k, int q = 0) {
    BfEdgeArgs a{};
    auto& es = h->es[q];
    a.snd = es.d_snd.as<int32_t>();
    a.rcv = es.d_rcv.as<int32_t>();
    a.E = h->g.set[q].e_local;
    a.ntiles = es.ntiles_e;
    a.P = es.bP.as<uint16_t>();
    a.Q = es.bQ.as<uint16_t>();
    a.Elat = es.bElat.as<uint16_t>();
    a.AGG = es.bAGG.as<uint16_t>();
    a.CARRY = es.bCARRY.as<uint16_t>();
    for (int i = 0; i < 3; ++i) a.chunk[i] = WB(h, h->bsoff[k].e_ch[q][i]);
    a.tabs = W(h, h->soff[k].e_tabs[q]);
    a.stamps = h->d_stamps.as<unsigned long long>();
    return a;
}

// project: the args feed k_project_bf16 (P,Q of set q); else k_node_bf16 (node MLP over all sets' aggregates)
BfNodeArgs bf_node_args(mgn_engine* h, int k, int q = 0, bool project = false) {
    BfNodeArgs a{};
    const auto& so = h->bsoff[k];
    a.n = h->g.n_own;
    a.ntiles = h->ntiles_n;
    a.rowptr = h->es[0].d_rowptr.as<int32_t>();
    a.V = h->bV.as<uint16_t>();
    a.AGG = h->es[0].bAGG.as<uint16_t>();
    a.CARRY = h->es[0].bCARRY.as<uint16_t>();
    a.P = h->es[q].bP.as<uint16_t>();
    a.Q = h->es[q].bQ.as<uint16_t>();
    for (int i = 0; i < 6; ++i) a.chunk[i] = WB(h, so.n_ch[i]);
    a.tabs = W(h, h->soff[k].n_tabs);
    if (q == 1) {
        a.chunk[4] = WB(h, so.p1_ch[0]);
        a.chunk[5] = WB(h, so.p1_ch[1]);
        a.tabs = W(h, h->soff[k].p1_tabs);
    }
    if (h->nsets > 1 && !project) {
        a.rowptr2 = h->es[1].d_rowptr.as<int32_t>();
        a.AGG2 = h->es[1].bAGG.as<uint16_t>();
        a.CARRY2 = h->es[1].bCARRY.as<uint16_t>();
        a.zero_row2 = 4 * tiles_or_one(h->es[1].ntiles_e);
        a.chunk[6] = WB(h, so.n_ch[6]);
    }
    a.zero_row = 4 * tiles_or_one(h->es[0].ntiles_e);
    a.tile0 = 0;
    return a;
}

inline bool is_bf16(const mgn_engine* h) { return h->cfg.dtype == MGN_BF16; }

}  // namespace

// =================================================================================================
extern "C" {

int mgn_abi_version(void) { return MGN_ABI_VERSION; }

int mgn_create(const mgn_config* cfg, mgn_handle** out) try {
    if (!out) return fail(nullptr, MGN_E_ARG, "null out pointer");
    *out = nullptr;
    // P / Q / CARRY rows are addressed through frag.hpp: prow_ptr in two translation units; objects compiled with different
    // MGN_PROW_BLOCK link without complaint and gather garbage
    if (kernels_prow_block() != split_prow_block())
        return fail(nullptr, MGN_E_STATE, "mgn_create: this library was linked from objects that disagree on the P / Q row layout (MGN_PROW_BLOCK %d in kernels.hip, %d in split.hip): rebuild all of it",
                    kernels_prow_block(), split_prow_block());
    std::string why;
    if (!cfg_ok(cfg, why)) return fail(nullptr, MGN_E_ARG, "mgn_create: %s", why.c_str());
    if (cfg->device == MGN_DEVICE_NONE) {
        mgn_engine* ho = new (std::nothrow) mgn_engine();
        if (!ho) return fail(nullptr, MGN_E_OOM, "host allocation failed");
        ho->cfg = *cfg;
        ho->host_only = true;
        layout_all(ho);
        *out = ho;
        return MGN_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, MGN_E_HIP, "mgn_create: no HIP device available (this engine has no CPU fallback)");
    if (cfg->device >= 0) {
        if (cfg->device >= ndev) return fail(nullptr, MGN_E_ARG, "mgn_create: device %d out of range (%d devices)", cfg->device, ndev);
        if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, MGN_E_HIP, "mgn_create: hipSetDevice failed");
    }
    mgn_engine* h = new (std::nothrow) mgn_engine();
    if (!h) return fail(nullptr, MGN_E_OOM, "host allocation failed");
    h->cfg = *cfg;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(nullptr, MGN_E_HIP, "mgn_create: hipStreamCreate failed");
    }
    h->stream = h->own_stream;
    if (const char* e = getenv("MGN_STAGGER_EDGE")) h->stagger_edge = atoi(e);
    if (const char* e = getenv("MGN_STAGGER_NODE")) h->stagger_node = atoi(e);
    if (const char* e = getenv("MGN_NODE_SPLIT")) h->node_split = atoi(e);
    if (const char* e = getenv("MGN_GRAPH")) h->use_graph = atoi(e);
    layout_all(h);
    *out = h;
    return MGN_OK;
} MGN_CATCH(nullptr)

void mgn_destroy(mgn_handle* h) {
    if (!h) return;
    if (h->host_only) { delete h->comm; delete h; return; }
    (void)hipStreamSynchronize(h->stream);
    drop_graph(h);
    for (auto& r : h->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
    delete h->comm;
    h->comm = nullptr;
    train_free(h);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

const char* mgn_last_error(const mgn_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mgn_set_stream(mgn_handle* h, void* hip_stream) try {
    if (int rc = need(h, false, false)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->stream = (hip_stream == MGN_STREAM_OWN) ? h->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
    return MGN_OK;
} MGN_CATCH(h)

int mgn_synchronize(mgn_handle* h) try {
    if (int rc = need(h, false, false)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MGN_OK;
} MGN_CATCH(h)

size_t mgn_param_count(const mgn_config* c) try {
    std::string why;
    if (!c || c->Fn < 1 || c->Fe < 1 || c->O < 1 || c->L < 1 || c->hidden_layers < 1 || c->hidden_layers > 4 || c->mps < 1) return 0;
    mgn_engine tmp;
    tmp.cfg = *c;
    return layout_all(&tmp);
} MGN_CATCH_SIZE

int mgn_set_params(mgn_handle* h, const float* packed, size_t n) try {
    if (!h || !packed) return fail(h, MGN_E_ARG, "mgn_set_params: null argument");
    if (int rc = need(h, false, false)) return rc;
    const size_t want = layout_all(h);
    if (n != want) return fail(h, MGN_E_ARG, "mgn_set_params: got %zu floats, model needs %zu", n, want);
    // the same values again (a caller that cannot tell whether its parameters changed calls this before every forward: 0.3 ms for the
    // comparison of 9 MB): nothing is invalidated, captured graphs and packed layouts stay
    if (h->have_params && h->params.size() == n && memcmp(h->params.data(), packed, n * sizeof(float)) == 0) return MGN_OK;
    h->params.assign(packed, packed + n);
    train_invalidate(h, 1);
    // The kernels' own weight layouts (three fp32 fragment orders, the bf16 pieces of the split path in two, the bf16 copies) take
    // ~30 ms of host time for the 15-step model; a training loop sets new parameters before EVERY step! and its kernels pack their own
    // (mgn_train.cpp) -- so they are built by the first call that reads them (need()), not here.
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    invalidate_static(h);
    h->packed_ok = false;
    h->have_params = true;
    return MGN_OK;
} MGN_CATCH(h)
}  // extern "C"

namespace mgn {
int pack_inference_weights(mgn_engine* h) {
    layout_all(h);
    const float* p = h->params.data();
    const mgn_config& c = h->cfg;
    const int L = c.L;
    const size_t CH = (size_t)L * L, TB = (size_t)T_COUNT * L;

    // The L x L chunks -- all but a megabyte of the ~60 MB of layouts -- are written ON THE DEVICE from the uploaded parameter vector
    // (train.hip: k_pack_weights; the host twins pack_chunk / pack_chunk_tmajor / pack_chunk16 / pack_chunk_bf16 / pack_chunk16_bf16
    // stay as their specification): the host describes them (WPackJob) and keeps packing the small things (tables, first-layer
    // vectors) into `f`, whose chunk regions stay empty and are not uploaded (segments between them are).  30 ms -> ~3 ms per
    // parameter change.
    std::vector<float> f;
    std::vector<WPackJob> jobs;
    std::vector<std::pair<size_t, size_t>> small;           // [begin, end) of what the host packed itself
    size_t seg0 = 0;
    // every chunk is stored three times: [lane-interleaved fragment order][t-major order (cooperative 32-row kernels)]
    // [16x16x4 fragment order (cooperative 16-row kernels, L = 128)]: copies of the chunk at offset `off` live at off + CH, off + 2 CH
    auto add_chunk = [&](const float* Wm, int ldw, int kbase) {
        const size_t off = f.size();
        if (off > seg0) small.push_back({seg0, off});
        f.resize(off + 3 * CH);
        seg0 = off + 3 * CH;
        jobs.push_back({0, (long long)off, (long long)(Wm - p), ldw, kbase, 1.f});
        return off;
    };
    auto add_tabs = [&](const float* b1, const float* b2, const float* b3, const float* ga, const float* be, const float* bq) {
        const size_t off = f.size();
        f.resize(off + 2 * TB);          // fragment order, then natural feature order (16-row kernels)
        const float* src[T_COUNT] = {b1, b2, b3, ga, be, bq, nullptr};
        for (int t = 0; t < T_COUNT; ++t) {
            pack_tab(f.data() + off + (size_t)t * L, src[t], L);
            for (int i = 0; i < L; ++i) f[off + TB + (size_t)t * L + i] = src[t] ? src[t][i] : 0.f;
        }
        for (int blk = 0; blk < 2; ++blk) {                    // T_LN: (eps_in, eps_out) of the LayerNorm variant (frag.hpp: ln_rstd)
            f[off + blk * TB + (size_t)T_LN * L + 0] = c.ln_mode == MGN_LN_STD_EPS ? 0.f : 1e-5f;
            f[off + blk * TB + (size_t)T_LN * L + 1] = c.ln_mode == MGN_LN_STD_EPS ? 1e-5f : 0.f;
        }
        return off;
    };
    auto add_w1f = [&](const float* W1, int K) {  // [K][L] -> per-k fragment tables
        const size_t off = f.size();
        f.resize(off + (size_t)K * L);
        for (int k = 0; k < K; ++k) pack_tab(f.data() + off + (size_t)k * L, W1 + (size_t)k * L, L);
        return off;
    };

    const int S = h->nsets;
    const MlpOff& e0 = h->es[0].pe[0];
    const int nl = c.hidden_layers + 1, nmid = nl - 2;      // Dense layers per MLP; L x L middle layers
    const bool h2 = c.hidden_layers == 2;                   // the tuned kernel families are specialised for this
    // the Dense layers after the first one of MLP m (GenOff): middle layers W[1 .. nmid], then the last one when it is L x L
    // (has_last: every MLP but the decoder), and their biases as tables.  Returns through g; ch[0] / ch[nmid] double as the
    // classic "W2" / "W3" slots of the tuned kernels at hidden_layers = 2.
    auto add_gen = [&](const MlpOff& m, bool has_last, GenOff& g) {
        for (int i = 0; i < nmid; ++i) g.ch[i] = add_chunk(p + m.W[1 + i], L, 0);
        if (has_last) g.ch[nmid] = add_chunk(p + m.W[nl - 1], L, 0);
        g.tabs = f.size();
        f.resize(f.size() + (size_t)(nmid + 1) * L);
        for (int i = 0; i < nmid; ++i) pack_tab(f.data() + g.tabs + (size_t)i * L, p + m.b[1 + i], L);
        pack_tab(f.data() + g.tabs + (size_t)nmid * L, has_last ? p + m.b[nl - 1] : nullptr, L);
    };
    auto b2 = [&](const MlpOff& m) { return h2 ? p + m.b[1] : nullptr; };
    auto b3 = [&](const MlpOff& m) { return h2 ? p + m.b[2] : nullptr; };
    // encoder, node side (+ projection onto step-0 edge-MLP layer 1 of set 0)
    {
        const MlpOff& m = h->enc_node;
        add_gen(m, true, h->en_gen);
        h->en_ch[0] = h->en_gen.ch[0];
        h->en_ch[1] = h->en_gen.ch[nmid];
        h->en_ch[2] = add_chunk(p + e0.W[0], L, 0);
        h->en_ch[3] = add_chunk(p + e0.W[0], L, L);
        h->en_tabs = add_tabs(p + m.b[0], b2(m), b3(m), p + m.gamma, p + m.beta, p + e0.b[0]);
        h->en_w1f = add_w1f(p + m.W[0], c.Fn);
    }
    for (int q = 0; q < S; ++q) {
        auto& es = h->es[q];
        const MlpOff& m = es.enc;
        add_gen(m, true, es.ee_gen);
        es.ee_ch[0] = es.ee_gen.ch[0];
        es.ee_ch[1] = es.ee_gen.ch[nmid];
        es.ee_tabs = add_tabs(p + m.b[0], b2(m), b3(m), p + m.gamma, p + m.beta, nullptr);
        es.ee_w1f = add_w1f(p + m.W[0], es.Fe);
    }
    h->soff.assign(c.mps, {});
    for (int k = 0; k < c.mps; ++k) {
        const MlpOff& mn = h->pn[k];
        const int kn = k + 1 < c.mps ? k + 1 : 0;                  // projection target (mode 2 at k=0 uses step 0 itself)
        auto& so = h->soff[k];
        for (int q = 0; q < S; ++q) {
            const MlpOff& me = h->es[q].pe[k];
            add_gen(me, true, so.e_gen[q]);
            so.e_ch[q][0] = so.e_gen[q].ch[0];
            so.e_ch[q][1] = so.e_gen[q].ch[nmid];
            so.e_ch[q][2] = add_chunk(p + me.W[0], L, 2 * L);
            so.e_tabs[q] = add_tabs(nullptr, b2(me), b3(me), p + me.gamma, p + me.beta, nullptr);
        }
        const MlpOff& nx = h->es[0].pe[kn];
        add_gen(mn, true, so.n_gen);
        so.n_ch[0] = so.n_gen.ch[0];
        so.n_ch[1] = so.n_gen.ch[nmid];
        so.n_ch[2] = add_chunk(p + mn.W[0], L, 0);
        so.n_ch[3] = add_chunk(p + mn.W[0], L, L);
        so.n_ch[4] = add_chunk(p + nx.W[0], L, 0);
        so.n_ch[5] = add_chunk(p + nx.W[0], L, L);
        so.n_tabs = add_tabs(p + mn.b[0], b2(mn), b3(mn), p + mn.gamma, p + mn.beta, p + nx.b[0]);
        if (S > 1) {
            const MlpOff& nx1 = h->es[1].pe[kn];
            so.n_ch[6] = add_chunk(p + mn.W[0], L, 2 * L);
            so.p1_ch[0] = add_chunk(p + nx1.W[0], L, 0);
            so.p1_ch[1] = add_chunk(p + nx1.W[0], L, L);
            so.p1_tabs = add_tabs(nullptr, nullptr, nullptr, nullptr, nullptr, p + nx1.b[0]);
        }
    }
    {
        const MlpOff& m = h->dec;
        h->de_ch[0] = add_chunk(p + m.W[0], L, 0);
        add_gen(m, false, h->de_gen);                                  // middle layers only: the last one (L -> O) runs on the VALU
        h->de_ch[1] = nmid > 0 ? h->de_gen.ch[0] : h->de_ch[0];
        h->de_tabs = add_tabs(p + m.b[0], b2(m), nullptr, nullptr, nullptr, nullptr);
        h->de_w3f = f.size();
        f.resize(f.size() + (size_t)c.O * L);
        for (int o = 0; o < c.O; ++o) pack_tab(f.data() + h->de_w3f + (size_t)o * L, p + m.W[nl - 1] + o, L, c.O);
        h->de_b3 = f.size();
        for (int o = 0; o < c.O; ++o) f.push_back(p[m.b[nl - 1] + o]);
        while (f.size() % 4) f.push_back(0.f);
    }
    // "project only" (mgn_proc_begin) needs step 0's own first layer in the projection slots of some NodeArgs: add a
    // dedicated pseudo-step at index mps (slots 0..3 alias step 0; tables carry bq = b1 of step 0).
    {
        mgn_engine::StepOff so = h->soff[0];
        so.n_ch[4] = h->en_ch[2];
        so.n_ch[5] = h->en_ch[3];
        so.n_tabs = add_tabs(nullptr, nullptr, nullptr, nullptr, nullptr, p + e0.b[0]);
        if (S > 1) {
            const MlpOff& e1 = h->es[1].pe[0];
            so.p1_ch[0] = add_chunk(p + e1.W[0], L, 0);
            so.p1_ch[1] = add_chunk(p + e1.W[0], L, L);
            so.p1_tabs = add_tabs(nullptr, nullptr, nullptr, nullptr, nullptr, p + e1.b[0]);
        }
        h->soff.push_back(so);
    }
    h->spoff.clear();
    if (L == 128 && c.hidden_layers == 2) {                            // bf16 pieces of the split path (split.hip): 4.4 MB per edge set,
        const bool node_side = S <= 2;                                  //   8.8 MB for the node side (+ 4.4 MB with a second edge set)
        // every chunk twice: the 32x32x16 fragment order (k_edge_ring, k_node_split, k_project_split; fp32 storage only) and the
        // 16x16x32 one (the 16-row cooperative kernels of small meshes -- in bf16 storage mode too, where they keep fp32-accurate
        // arithmetic --)
        const bool f32 = c.dtype == MGN_F32;
        h->spoff.assign(c.mps + 1, {});
        size_t off = 0;
        auto put = [&](const float* src, int kb, size_t& o32, size_t& o16) {
            o32 = 0;
            if (f32) {
                jobs.push_back({1, (long long)off, (long long)(src - p), L, kb, 1.f});
                o32 = off;
                off += (size_t)3 * 16384;
            }
            jobs.push_back({2, (long long)off, (long long)(src - p), L, kb, 1.f});
            o16 = off;
            off += (size_t)3 * 16384;
        };
        // two fp16 pieces of the same chunk times a power of two that puts its largest entry into [2^14, 2^15) (split_common.hpp): in the
        // 32x32x16 fragment order (fp32 storage: k_edge_ring_h, k_node_split_h, k_project_split_h) and in the 16x16x32 one (16-row kernels)
        auto puth = [&](const float* src, int kb, size_t& oh, size_t& o16h, float& sc) {
            float mx = 0.f;
            for (int k = 0; k < L; ++k)
                for (int n = 0; n < L; ++n) mx = std::max(mx, std::fabs(src[(size_t)(kb + k) * L + n]));
            int e = 0;
            (void)std::frexp(mx, &e);                                   // mx = m 2^e, m in [0.5, 1): floor(log2 mx) = e - 1
            if (!(mx > 0.f) || e - 1 < -40) e = -39;
            if (!std::isfinite(mx)) e = 128;
            sc = std::ldexp(1.f, 15 - e);
            oh = 0;
            if (f32) {
                jobs.push_back({4, (long long)off, (long long)(src - p), L, kb, sc});
                oh = off;
                off += (size_t)2 * 16384;
            }
            jobs.push_back({5, (long long)off, (long long)(src - p), L, kb, sc});
            o16h = off;
            off += (size_t)2 * 16384;
        };
        for (int k = 0; k < c.mps; ++k)
            for (int q = 0; q < S; ++q) {
                const MlpOff& me = h->es[q].pe[k];
                const float* src[3] = {p + me.W[1], p + me.W[2], p + me.W[0]};
                const int kb[3] = {0, 0, 2 * L};
                for (int i = 0; i < 3; ++i) put(src[i], kb[i], h->spoff[k].e_ch[q][i], h->spoff[k].e16_ch[q][i]);
                for (int i = 0; i < 3; ++i) puth(src[i], kb[i], h->spoff[k].eh_ch[q][i], h->spoff[k].e16h_ch[q][i], h->spoff[k].eh_s[q][i]);
                float bp = 0.f;
                for (int i = 0; i < L; ++i) bp = std::max(bp, p[me.b[1] + i]);
                h->spoff[k].e_b2pos[q] = bp;
                h->spoff[k].have_h = true;
            }
        for (int k = 0; node_side && k <= c.mps; ++k) {                 // node MLP of step k + projection for step k + 1 (k = mps: the
            const MlpOff& mn = h->pn[k < c.mps ? k : 0];                //   "project only" pseudo-step: step 0's own first layer)
            const MlpOff& nx = h->es[0].pe[k + 1 < c.mps ? k + 1 : 0];
            const float* src[6] = {p + mn.W[1], p + mn.W[2], p + mn.W[0], p + mn.W[0], p + nx.W[0], p + nx.W[0]};
            const int kb[6] = {0, 0, 0, L, 0, L};
            for (int i = 0; i < 6; ++i) put(src[i], kb[i], h->spoff[k].n_ch[i], h->spoff[k].n16_ch[i]);
            for (int i = 0; i < 6; ++i) puth(src[i], kb[i], h->spoff[k].nh_ch[i], h->spoff[k].n16h_ch[i], h->spoff[k].nh_s[i]);
            float bp = 0.f;
            for (int i = 0; i < L; ++i) bp = std::max(bp, p[mn.b[1] + i]);
            h->spoff[k].n_b2pos = bp;
            h->spoff[k].have_h = true;
            if (S == 2) {                                               // second edge set: its aggregate block of the node MLP, its projection
                const MlpOff& n1 = h->es[1].pe[k + 1 < c.mps ? k + 1 : 0];
                const float* src2[3] = {p + mn.W[0], p + n1.W[0], p + n1.W[0]};
                const int kb2[3] = {2 * L, 0, L};
                for (int i = 0; i < 3; ++i) put(src2[i], kb2[i], h->spoff[k].n2_ch[i], h->spoff[k].n16_ch[6 + i]);
                size_t unused = 0;
                for (int i = 0; i < 3; ++i) puth(src2[i], kb2[i], unused, h->spoff[k].n16h_ch[6 + i], h->spoff[k].nh_s[6 + i]);
            }
            h->spoff[k].have_n = true;
        }
        h->have_ench = false;
        if (f32) {          // encoders and decoder (32x32x16 order only: their kernels are the 32-row ones)
            size_t unused = 0;
            auto only32 = [&](const float* src, int kb, size_t& oh, float& sc) {
                const size_t before = jobs.size();
                puth(src, kb, oh, unused, sc);
                jobs.pop_back();                                          // (drop the 16x16x32 copy puth appended)
                off -= (size_t)2 * 16384;
                (void)before;
            };
            const MlpOff& mn = h->enc_node;
            only32(p + mn.W[1], 0, h->enh_ch[0], h->enh_s[0]);
            only32(p + mn.W[2], 0, h->enh_ch[1], h->enh_s[1]);
            only32(p + e0.W[0], 0, h->enh_ch[2], h->enh_s[2]);
            only32(p + e0.W[0], L, h->enh_ch[3], h->enh_s[3]);
            for (int q = 0; q < S; ++q) {
                const MlpOff& me = h->es[q].enc;
                only32(p + me.W[1], 0, h->eeh_ch[q][0], h->eeh_s[q][0]);
                only32(p + me.W[2], 0, h->eeh_ch[q][1], h->eeh_s[q][1]);
            }
            const MlpOff& md = h->dec;
            only32(p + md.W[0], 0, h->deh_ch[0], h->deh_s[0]);
            only32(p + md.W[1], 0, h->deh_ch[1], h->deh_s[1]);
            h->have_ench = true;
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, h->wsp.ensure(off * 2));
    }
    size_t wb_size = 0;
    if (c.dtype == MGN_BF16) {
        auto addb = [&](const float* Wm, int kbase) {
            const size_t off = wb_size;
            wb_size += (size_t)L * L;
            jobs.push_back({3, (long long)off, (long long)(Wm - p), L, kbase, 1.f});
            return off;
        };
        h->bsoff.assign(c.mps + 1, {});
        for (int k = 0; k < c.mps; ++k) {
            const MlpOff& mn = h->pn[k];
            const int kn = k + 1 < c.mps ? k + 1 : 0;
            const MlpOff& nx = h->es[0].pe[kn];
            auto& so = h->bsoff[k];
            for (int q = 0; q < S; ++q) {
                const MlpOff& me = h->es[q].pe[k];
                so.e_ch[q][0] = addb(p + me.W[1], 0);
                so.e_ch[q][1] = addb(p + me.W[2], 0);
                so.e_ch[q][2] = addb(p + me.W[0], 2 * L);
            }
            so.n_ch[0] = addb(p + mn.W[1], 0);
            so.n_ch[1] = addb(p + mn.W[2], 0);
            so.n_ch[2] = addb(p + mn.W[0], 0);
            so.n_ch[3] = addb(p + mn.W[0], L);
            so.n_ch[4] = addb(p + nx.W[0], 0);
            so.n_ch[5] = addb(p + nx.W[0], L);
            if (S > 1) {
                const MlpOff& nx1 = h->es[1].pe[kn];
                so.n_ch[6] = addb(p + mn.W[0], 2 * L);
                so.p1_ch[0] = addb(p + nx1.W[0], 0);
                so.p1_ch[1] = addb(p + nx1.W[0], L);
            }
        }
        h->bsoff[c.mps] = h->bsoff[0];                       // projection for step 0 (mgn_proc_begin)
        h->bsoff[c.mps].n_ch[4] = addb(p + e0.W[0], 0);
        h->bsoff[c.mps].n_ch[5] = addb(p + e0.W[0], L);
        if (S > 1) {
            const MlpOff& e1 = h->es[1].pe[0];
            h->bsoff[c.mps].p1_ch[0] = addb(p + e1.W[0], 0);
            h->bsoff[c.mps].p1_ch[1] = addb(p + e1.W[0], L);
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, h->wbf.ensure(wb_size * 2));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    invalidate_static(h);
    if (f.size() > seg0) small.push_back({seg0, f.size()});
    HIPCHK(h, h->wfrag.ensure(f.size() * 4));
    HIPCHK(h, h->d_params.ensure(h->params.size() * 4));
    HIPCHK(h, h->d_wjobs.ensure(jobs.size() * sizeof(WPackJob)));
    HIPCHK(h, hipMemcpyAsync(h->d_params.p, p, h->params.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_wjobs.p, jobs.data(), jobs.size() * sizeof(WPackJob), hipMemcpyHostToDevice, h->stream));
    for (const auto& sg : small)
        HIPCHK(h, hipMemcpyAsync(h->wfrag.as<float>() + sg.first, f.data() + sg.first, (sg.second - sg.first) * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_pack_weights(L, h->d_wjobs.as<WPackJob>(), (int)jobs.size(), h->d_params.as<float>(), h->wfrag.as<float>(),
                                  h->wsp.as<uint16_t>(), h->wbf.as<uint16_t>(), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));       // (f and jobs are locals: their copies must have left the host)
    h->wjobs = std::move(jobs);
    h->packed_ok = true;
    return MGN_OK;
}
}  // namespace mgn
extern "C" {

int mgn_get_params(mgn_handle* h, float* packed, size_t n) try {
    if (int rc = need(h, true, false, false)) return rc;
    if (!packed || n != h->params.size()) return fail(h, MGN_E_ARG, "mgn_get_params: size mismatch");
    memcpy(packed, h->params.data(), n * sizeof(float));
    return MGN_OK;
} MGN_CATCH(h)

int mgn_set_norms(mgn_handle* h, const float* ns, const float* nsh, const float* es, const float* esh, const float* os,
                  const float* osh) try {
    if (int rc = need(h, false, false)) return rc;
    const mgn_config& c = h->cfg;
    if ((ns == nullptr) != (nsh == nullptr) || (es == nullptr) != (esh == nullptr) || (os == nullptr) != (osh == nullptr))
        return fail(h, MGN_E_ARG, "mgn_set_norms: scale and shift must both be given or both be NULL");
    std::vector<float> v((size_t)2 * (c.Fn + c.Fe + c.O), 0.f);
    float* q = v.data();
    auto put = [&](const float* a, const float* b, int n) {
        for (int i = 0; i < n; ++i) { q[i] = a ? a[i] : 1.f; q[n + i] = b ? b[i] : 0.f; }
        q += 2 * n;
    };
    put(ns, nsh, c.Fn);
    put(es, esh, c.Fe);
    put(os, osh, c.O);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, h->norms.ensure(v.size() * 4));
    HIPCHK(h, hipMemcpy(h->norms.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    h->norms_host = v;
    invalidate_static(h);
    h->have_nnorm = ns != nullptr;
    h->have_enorm = es != nullptr;
    h->have_onorm = os != nullptr;
    return MGN_OK;
} MGN_CATCH(h)

// Node numbering policy of mgn_set_graph (graph_host.h: build_local_graph): 1 = keep the caller's numbering unless the breadth-first
// order of the mesh is RENUMBER_GAIN times more local (DeepMind's trajectories carry arbitrary node numbers and create_base_graph
// passes them through, reference src/graph.jl:30-36; the processor kernels gather sender rows and lose 4-9 % (fp32) / 25 % (bf16)
// on a scattered numbering), 0 = never, 2 = always.  Invisible at the boundary: every array crosses it in the caller's order.
static int g_renumber = [] { const char* e = getenv("MGN_RENUMBER"); return e ? atoi(e) : 1; }();

// (re)build the local graph from the kept global edge lists and upload it.  keep_owner: node partition unchanged
static int rebuild_graph(mgn_handle* h, int32_t N, const EdgeList* sets, const float* mesh_pos, int32_t pos_dim, bool keep_owner,
                         const char* who, const int32_t* owner_in = nullptr) {
    h->have_graph = false;
    h->hx_ready = false;
    h->all_gid.clear();
    invalidate_static(h);
    train_invalidate(h, 2);
    if (!h->host_only) { (void)hipStreamSynchronize(h->stream); drop_graph(h); }
    const std::string why = build_local_graph(N, h->nsets, sets, mesh_pos, pos_dim, owner_in ? owner_in : (keep_owner ? h->g.owner.data() : nullptr),
                                              h->cfg.rank, h->cfg.nranks, h->g,
                                              g_renumber);   // (the numbering follows set 0 alone: a later mgn_set_edge_set / mgn_world_edges_dev keeps it)
    if (!why.empty()) return fail(h, MGN_E_ARG, "%s: %s", who, why.c_str());
    const LocalGraph& g = h->g;
    for (int q = 0; q < h->nsets; ++q) h->es[q].ntiles_e = (int32_t)((g.set[q].e_local + TILE - 1) / TILE);
    h->ntiles_n = (g.n_own + TILE - 1) / TILE;
    if (h->host_only) {
        h->have_graph = true;
        return MGN_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    auto up = [&](DevBuf& d, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = d.ensure(bytes);
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice);
    };
    for (int q = 0; q < h->nsets; ++q) {
        const EdgeTopo& t = g.set[q];
        auto& es = h->es[q];
        HIPCHK(h, up(es.d_snd, t.snd.data(), t.snd.size() * 4));
        HIPCHK(h, up(es.d_rcv, t.rcv.data(), t.rcv.size() * 4));
        HIPCHK(h, up(es.d_rowptr, t.rowptr.data(), t.rowptr.size() * 4));
        HIPCHK(h, up(es.d_edge_gid, t.edge_gid.data(), t.edge_gid.size() * 8));
    }
    HIPCHK(h, up(h->d_own_gid, g.own_gid.data(), g.own_gid.size() * 4));
    HIPCHK(h, up(h->d_send_idx, g.send_idx.data(), g.send_idx.size() * 4));
    return MGN_OK;
}

int mgn_set_graph(mgn_handle* h, int32_t N, int64_t E, const int32_t* senders, const int32_t* receivers, int32_t index_base,
                  const float* mesh_pos, int32_t pos_dim) try {
    if (!h) return MGN_E_ARG;
    if (index_base != 0 && index_base != 1) return fail(h, MGN_E_ARG, "mgn_set_graph: index_base must be 0 or 1");
    if (E < 0 || (E > 0 && (!senders || !receivers))) return fail(h, MGN_E_ARG, "mgn_set_graph: null senders/receivers");
    EdgeList sets[MAX_EDGE_SETS];
    sets[0] = {E, senders, receivers, index_base};
    if (h->nsets > 1) {   // the second set starts empty; keep set 0's global list for the rebuild in mgn_set_edge_set
        h->es[0].gs.assign(senders, senders + E);
        h->es[0].gr.assign(receivers, receivers + E);
        h->es[0].gbase = index_base;
        h->es[1].gs.clear();
        h->es[1].gr.clear();
        h->es[1].have_ef = false;
    }
    if (int rc = rebuild_graph(h, N, sets, mesh_pos, pos_dim, false, "mgn_set_graph")) return rc;
    if (h->host_only) return MGN_OK;
    if (int rc = alloc_latents(h)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_graph = true;
    return MGN_OK;
} MGN_CATCH(h)

// Rank-local ingest (SURVEY.md 8e at scale): what mgn_set_graph computes from the global lists, from the rank's own part of them.
int mgn_partition_nodes(int32_t N, const float* mesh_pos, int32_t pos_dim, int32_t nranks, int32_t* owner) try {
    if (N < 0 || nranks < 1 || (N > 0 && !owner) || (mesh_pos && pos_dim <= 0)) return MGN_E_ARG;
    std::vector<int32_t> o;
    rcb_partition(N, mesh_pos, pos_dim, nranks, o);
    if (N > 0) memcpy(owner, o.data(), (size_t)N * 4);
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_set_graph_local(mgn_handle* h, int32_t N, const int32_t* owner, int64_t E_global, int64_t E_touch, const int32_t* senders,
                        const int32_t* receivers, const int64_t* edge_gid, int32_t index_base) try {
    if (!h) return MGN_E_ARG;
    if (h->nsets != 1) return fail(h, MGN_E_UNSUPPORTED, "mgn_set_graph_local: one edge set (a second set is rebuilt from the global lists)");
    if (index_base != 0 && index_base != 1) return fail(h, MGN_E_ARG, "mgn_set_graph_local: index_base must be 0 or 1");
    if (N < 0 || (N > 0 && !owner)) return fail(h, MGN_E_ARG, "mgn_set_graph_local: null owner map");
    if (E_touch < 0 || E_global < E_touch || (E_touch > 0 && (!senders || !receivers || !edge_gid)))
        return fail(h, MGN_E_ARG, "mgn_set_graph_local: null or inconsistent edge arrays");
    for (int32_t i = 0; i < N; ++i)
        if (owner[i] < 0 || owner[i] >= h->cfg.nranks) return fail(h, MGN_E_ARG, "mgn_set_graph_local: owner[%d] = %d outside [0, nranks)", i, owner[i]);
    EdgeList sets[MAX_EDGE_SETS];
    sets[0] = {E_touch, senders, receivers, index_base, edge_gid, E_global};
    if (int rc = rebuild_graph(h, N, sets, nullptr, 0, false, "mgn_set_graph_local", owner)) return rc;
    if (h->host_only) return MGN_OK;
    if (int rc = alloc_latents(h)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_graph = true;
    return MGN_OK;
} MGN_CATCH(h)

int mgn_set_edge_set(mgn_handle* h, int32_t set, int64_t E, const int32_t* senders, const int32_t* receivers, int32_t index_base) try {
    if (!h) return MGN_E_ARG;
    if (!h->have_graph) return fail(h, MGN_E_STATE, "mgn_set_edge_set before mgn_set_graph");
    if (set < 1 || set >= h->nsets) return fail(h, MGN_E_ARG, "mgn_set_edge_set: set %d out of range (handle has %d edge sets; set 0 is mgn_set_graph's)", set, h->nsets);
    if (index_base != 0 && index_base != 1) return fail(h, MGN_E_ARG, "mgn_set_edge_set: index_base must be 0 or 1");
    if (E < 0 || (E > 0 && (!senders || !receivers))) return fail(h, MGN_E_ARG, "mgn_set_edge_set: null senders/receivers");
    auto& e1 = h->es[set];
    e1.gs.assign(senders, senders + E);
    e1.gr.assign(receivers, receivers + E);
    e1.gbase = index_base;
    e1.have_ef = false;
    EdgeList sets[MAX_EDGE_SETS];
    for (int q = 0; q < h->nsets; ++q) sets[q] = {(int64_t)h->es[q].gs.size(), h->es[q].gs.data(), h->es[q].gr.data(), h->es[q].gbase};
    const int32_t n_halo_before = h->g.n_halo;
    if (int rc = rebuild_graph(h, h->g.N, sets, nullptr, 0, true, "mgn_set_edge_set")) return rc;
    if (h->host_only) return MGN_OK;
    // node numbering is unchanged with one partition (no boundary / halo): V and set 0's latents stay valid and only the
    // new set's buffers are (re)sized; a changed halo re-sizes everything that is indexed by owned + halo rows
    if (h->cfg.nranks == 1 && n_halo_before == 0) {
        if (int rc = alloc_edge_set(h, set)) return rc;
    } else if (int rc = alloc_latents(h)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_graph = true;
    return MGN_OK;
} MGN_CATCH(h)

int mgn_edge_set_info(const mgn_handle* h, int32_t set, int64_t* E, int64_t* e_local) try {
    if (!h || !h->have_graph) return MGN_E_STATE;
    if (set < 0 || set >= h->nsets) return MGN_E_ARG;
    if (E) *E = h->g.set[set].E;
    if (e_local) *e_local = h->g.set[set].e_local;
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_set_edge_features(mgn_handle* h, int32_t set, const float* ef) try {
    if (int rc = need(h, false, true)) return rc;
    if (set < 1 || set >= h->nsets) return fail(h, MGN_E_ARG, "mgn_set_edge_features: set %d out of range", set);
    auto& es = h->es[set];
    const size_t bytes = (size_t)h->g.set[set].E * es.Fe * 4;
    if (bytes && !ef) return fail(h, MGN_E_ARG, "mgn_set_edge_features: null input");
    const void* before = es.d_ef.p;
    HIPCHK(h, es.d_ef.ensure(bytes));
    if (es.d_ef.p != before && h->fwd_exec) {   // the captured forward reads this buffer
        (void)hipStreamSynchronize(h->stream);
        (void)hipGraphExecDestroy(h->fwd_exec);
        h->fwd_exec = nullptr;
        h->fwd_warm = false;
    }
    if (bytes) HIPCHK(h, hipMemcpyAsync(es.d_ef.p, ef, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    es.have_ef = true;
    return MGN_OK;
} MGN_CATCH(h)

int mgn_partition_info(const mgn_handle* h, int32_t* n_own, int32_t* n_halo, int64_t* e_local) try {
    if (!h || !h->have_graph) return MGN_E_STATE;
    if (n_own) *n_own = h->g.n_own;
    if (n_halo) *n_halo = h->g.n_halo;
    if (e_local) *e_local = h->g.set[0].e_local;
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_owned_nodes(const mgn_handle* h, int32_t* ids) try {
    if (!h || !h->have_graph || !ids) return MGN_E_STATE;
    memcpy(ids, h->g.own_gid.data(), h->g.own_gid.size() * 4);
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_local_edges(const mgn_handle* h, int64_t* ids) try {
    if (!h || !h->have_graph || !ids) return MGN_E_STATE;
    memcpy(ids, h->g.set[0].edge_gid.data(), h->g.set[0].edge_gid.size() * 8);
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_halo_counts(const mgn_handle* h, int32_t* send_rows, int32_t* recv_rows) try {
    if (!h || !h->have_graph) return MGN_E_STATE;
    for (int q = 0; q < h->cfg.nranks; ++q) {
        if (send_rows) send_rows[q] = h->g.send_rows[q];
        if (recv_rows) recv_rows[q] = h->g.recv_rows[q];
    }
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_halo_nodes(const mgn_handle* h, int32_t* ids) try {
    if (!h || !h->have_graph || !ids) return MGN_E_STATE;
    memcpy(ids, h->g.halo_gid.data(), h->g.halo_gid.size() * 4);
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_halo_send_index(const mgn_handle* h, int32_t* rows) try {
    if (!h || !h->have_graph || !rows) return MGN_E_STATE;
    memcpy(rows, h->g.send_idx.data(), h->g.send_idx.size() * 4);
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_local_graph(const mgn_handle* h, int32_t* snd, int32_t* rcv, int32_t* rowptr) try {
    if (!h || !h->have_graph) return MGN_E_STATE;
    if (snd) memcpy(snd, h->g.set[0].snd.data(), h->g.set[0].snd.size() * 4);
    if (rcv) memcpy(rcv, h->g.set[0].rcv.data(), h->g.set[0].rcv.size() * 4);
    if (rowptr) memcpy(rowptr, h->g.set[0].rowptr.data(), h->g.set[0].rowptr.size() * 4);
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_node_owner(const mgn_handle* h, int32_t* owner) try {
    if (!h || !h->have_graph || !owner) return MGN_E_STATE;
    memcpy(owner, h->g.owner.data(), h->g.owner.size() * 4);
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_boundary_count(const mgn_handle* h, int32_t* n_boundary) try {
    if (!h || !h->have_graph || !n_boundary) return MGN_E_STATE;
    *n_boundary = h->g.n_boundary;
    return MGN_OK;
} MGN_CATCH(nullptr)

// ---- staged pipeline ----------------------------------------------------------------------------
// engine_order: the node rows must sit on the device in the ENGINE's order even on one partition -- the callers that later overwrite
// the state slot of d_nfA with rows in that order (mgn_set_static + mgn_ode_step(x), mgn_rollout)
static int upload_inputs(mgn_handle* h, const float* a, int wa, const float* b, int wb, const float* ef, bool engine_order = false) {
    const LocalGraph& g = h->g;
    h->in_wa = wa;
    h->in_wb = wb;
    if (h->cfg.nranks > 1 || (engine_order && g.renumbered)) {
        // a partition needs 1 / nranks of the inputs: gather the owned node rows and the local edge rows on the host and
        // upload those (M-1M on 8 GPUs: 13 MB instead of 108 MB per rank and forward); the encoders then read them in
        // local order (null gid).  (A renumbered SINGLE partition, graph_host.h, uploads the caller's arrays as they are for
        // mgn_forward and the one-shot mgn_ode_step -- one contiguous copy each -- and the encoders gather through own_gid / edge_gid on
        // the device: a host gather of 6 M edge rows costs tens of ms per forward.  Where the state is kept in the engine's order
        // (engine_order), it goes the partitions' way.)
        h->in_local = true;
        const EdgeTopo& t = g.set[0];
        const int Fe = h->cfg.Fe;
        std::vector<float> loc((size_t)g.n_own * (wa + wb) + (size_t)t.e_local * Fe);
        float* la = loc.data();
        float* lb = la + (size_t)g.n_own * wa;
        float* le = lb + (size_t)g.n_own * wb;
        for (int32_t i = 0; i < g.n_own; ++i) {
            memcpy(la + (size_t)i * wa, a + (size_t)g.own_gid[i] * wa, (size_t)wa * 4);
            if (wb > 0) memcpy(lb + (size_t)i * wb, b + (size_t)g.own_gid[i] * wb, (size_t)wb * 4);
        }
        for (int64_t j = 0; j < t.e_local; ++j) memcpy(le + (size_t)j * Fe, ef + (size_t)t.edge_gid[j] * Fe, (size_t)Fe * 4);
        HIPCHK(h, h->d_nfA.ensure((size_t)g.n_own * wa * 4));
        HIPCHK(h, h->d_nfB.ensure((size_t)g.n_own * (wb > 0 ? wb : 1) * 4));
        HIPCHK(h, h->es[0].d_ef.ensure((size_t)t.e_local * Fe * 4));
        HIPCHK(h, hipMemcpyAsync(h->d_nfA.p, la, (size_t)g.n_own * wa * 4, hipMemcpyHostToDevice, h->stream));
        if (wb > 0) HIPCHK(h, hipMemcpyAsync(h->d_nfB.p, lb, (size_t)g.n_own * wb * 4, hipMemcpyHostToDevice, h->stream));
        if (t.e_local > 0) HIPCHK(h, hipMemcpyAsync(h->es[0].d_ef.p, le, (size_t)t.e_local * Fe * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));   // `loc` goes out of scope
        return MGN_OK;
    }
    h->in_local = false;
    HIPCHK(h, h->d_nfA.ensure((size_t)g.N * wa * 4));
    HIPCHK(h, hipMemcpyAsync(h->d_nfA.p, a, (size_t)g.N * wa * 4, hipMemcpyHostToDevice, h->stream));
    if (wb > 0) {
        HIPCHK(h, h->d_nfB.ensure((size_t)g.N * wb * 4));
        HIPCHK(h, hipMemcpyAsync(h->d_nfB.p, b, (size_t)g.N * wb * 4, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, h->es[0].d_ef.ensure((size_t)g.set[0].E * h->cfg.Fe * 4));
    HIPCHK(h, hipMemcpyAsync(h->es[0].d_ef.p, ef, (size_t)g.set[0].E * h->cfg.Fe * 4, hipMemcpyHostToDevice, h->stream));
    return MGN_OK;
}

int mgn_fwd_upload(mgn_handle* h, const float* nf, const float* ef) try {
    if (int rc = need(h, false, true)) return rc;
    invalidate_static(h);
    if (!nf || (!ef && h->g.set[0].E > 0)) return fail(h, MGN_E_ARG, "mgn_fwd_upload: null input");
    return upload_inputs(h, nf, h->cfg.Fn, nullptr, 0, ef);
} MGN_CATCH(h)

static int project_set(mgn_handle* h, int k, int q, int32_t tile0 = 0, int32_t ntiles = -1) {   // P,Q of set q for step k (k = mps: step 0)
    if (ntiles < 0) ntiles = h->ntiles_n;
    if (is_bf16(h) && !use_c16(h)) {
        BfNodeArgs b = bf_node_args(h, k, q, true);
        b.tile0 = tile0;
        b.ntiles = ntiles;
        HIPCHK(h, launch_project_bf16(b, h->stream));
        return MGN_OK;
    }
    NodeArgs a = node_args(h, k, 2, q);
    a.tile0 = tile0;
    a.ntiles = ntiles;
    HIPCHK(h, launch_project(h->cfg.L, a, h->stream));
    return MGN_OK;
}

static int encode_impl(mgn_handle* h, bool use_norms, bool nodes = true, bool edges = true) {
    const mgn_config& c = h->cfg;
    const LocalGraph& g = h->g;
    const float* nrm = h->norms.as<float>();
    ProfScope ps(h, F_ENC);
    if (nodes) {
        EncNodeArgs a{};
        a.n = g.n_own;
        a.ntiles = h->ntiles_n;
        a.gid = h->in_local ? nullptr : h->d_own_gid.as<int32_t>();
        a.srcA = h->srcA_override ? h->srcA_override : h->d_nfA.as<float>();
        a.wa = h->in_wa;
        a.srcB = h->d_nfB.as<float>();
        a.wb = h->in_wb;
        if (use_norms && h->have_nnorm) { a.scale = nrm; a.shift = nrm + c.Fn; }
        a.w1f = W(h, h->en_w1f);
        a.V = h->V.as<float>();
        a.P = h->es[0].P.as<float>();
        a.Q = h->es[0].Q.as<float>();
        for (int i = 0; i < 4; ++i) a.chunk[i] = W(h, h->en_ch[i]);
        for (int i = 0; i < 4; ++i) {
            a.splith[i] = (h->have_ench && h->wsp.p) ? h->wsp.as<uint16_t>() + h->enh_ch[i] : nullptr;
            a.h2_rs[i] = h->have_ench ? 1.f / h->enh_s[i] : 1.f;
        }
        a.tabs = W(h, h->en_tabs);
        a.gen = gen_of(h, h->en_gen, true);
        HIPCHK(h, launch_enc_node(c.L, a, h->stream));
        if (is_bf16(h)) {   // fp32 encoder output -> bf16 state; P,Q of step 0 from the bf16 latents
            HIPCHK(h, launch_tile_f32_to_bf16(h->V.as<float>(), h->bV.as<uint16_t>(), h->ntiles_n, h->stream));
            if (int rc = project_set(h, c.mps, 0)) return rc;
        }
        for (int q = 1; q < h->nsets; ++q)
            if (int rc = project_set(h, c.mps, q)) return rc;
    }
    if (edges) {
        for (int q = 0; q < h->nsets; ++q) {
            auto& es = h->es[q];
            EncEdgeArgs b{};
            b.E = g.set[q].e_local;
            b.ntiles = es.ntiles_e;
            b.gid = (q == 0 && h->in_local) ? nullptr : es.d_edge_gid.as<int64_t>();
            b.ef = es.d_ef.as<float>();
            b.Fe = es.Fe;
            if (q == 0 && use_norms && h->have_enorm) { b.scale = nrm + 2 * c.Fn; b.shift = nrm + 2 * c.Fn + c.Fe; }
            b.w1f = W(h, es.ee_w1f);
            b.Elat = es.Elat.as<float>();
            for (int i = 0; i < 2; ++i) b.chunk[i] = W(h, es.ee_ch[i]);
            for (int i = 0; i < 2; ++i) {
                b.splith[i] = (h->have_ench && h->wsp.p) ? h->wsp.as<uint16_t>() + h->eeh_ch[q][i] : nullptr;
                b.h2_rs[i] = h->have_ench ? 1.f / h->eeh_s[q][i] : 1.f;
            }
            b.tabs = W(h, es.ee_tabs);
            b.gen = gen_of(h, es.ee_gen, true);
            HIPCHK(h, launch_enc_edge(c.L, b, h->stream));
            if (is_bf16(h)) HIPCHK(h, launch_tile_f32_to_bf16(es.Elat.as<float>(), es.bElat.as<uint16_t>(), es.ntiles_e, h->stream));
        }
    }
    return MGN_OK;
}

static int need_set_features(mgn_handle* h) {
    for (int q = 1; q < h->nsets; ++q)
        if (h->g.set[q].E > 0 && !h->es[q].have_ef)
            return fail(h, MGN_E_STATE, "edge set %d has edges but no features: call mgn_set_edge_features after mgn_set_edge_set", q);
    return MGN_OK;
}

int mgn_fwd_encode(mgn_handle* h) try {
    if (int rc = need(h, true, true)) return rc;
    if (!h->d_nfA.p) return fail(h, MGN_E_STATE, "mgn_fwd_encode before mgn_fwd_upload");
    if (int rc = need_set_features(h)) return rc;
    return encode_impl(h, false);
} MGN_CATCH(h)

int mgn_proc_begin(mgn_handle* h) try {
    if (int rc = need(h, true, true)) return rc;
    ProfScope ps(h, F_NODE);
    for (int q = 0; q < h->nsets; ++q)
        if (int rc = project_set(h, h->cfg.mps, q)) return rc;
    return MGN_OK;
} MGN_CATCH(h)

static int proc_edge_range(mgn_handle* h, int32_t k, int32_t phase) {   // phase 0: all tiles, 1: interior tiles, 2: boundary tiles
    bool work = false;
    for (int q = 0; q < h->nsets; ++q) {
        const int32_t tb = boundary_tiles(h, q), nt = h->es[q].ntiles_e;
        work = work || (phase == 0 ? nt : (phase == 1 ? nt - tb : tb)) > 0;
    }
    if (!work) return MGN_OK;
    ProfScope ps(h, phase == 2 ? F_EDGE_BND : F_EDGE);
    for (int q = 0; q < h->nsets; ++q) {
        const int32_t tb = boundary_tiles(h, q), nt = h->es[q].ntiles_e;
        const int32_t t0 = phase == 1 ? tb : 0, n = phase == 0 ? nt : (phase == 1 ? nt - tb : tb);
        if (n <= 0) continue;
        if (is_bf16(h) && !use_c16(h)) {
            BfEdgeArgs a = bf_edge_args(h, k, q);
            a.tile0 = t0;
            a.ntiles = n;
            HIPCHK(h, launch_edge_bf16(a, h->stream));
        } else {
            EdgeArgs a = edge_args(h, k, q);
            a.tile0 = t0;
            a.ntiles = n;
            HIPCHK(h, launch_edge_step(h->cfg.L, a, h->stream));
        }
    }
    return MGN_OK;
}

int mgn_proc_edge(mgn_handle* h, int32_t k) try {
    if (int rc = need(h, true, true)) return rc;
    if (k < 0 || k >= h->cfg.mps) return fail(h, MGN_E_ARG, "mgn_proc_edge: step %d out of range", k);
    return proc_edge_range(h, k, 0);
} MGN_CATCH(h)

int mgn_proc_edge_phase(mgn_handle* h, int32_t k, int32_t phase) try {
    if (int rc = need(h, true, true)) return rc;
    if (k < 0 || k >= h->cfg.mps) return fail(h, MGN_E_ARG, "mgn_proc_edge_phase: step %d out of range", k);
    if (phase != 1 && phase != 2) return fail(h, MGN_E_ARG, "mgn_proc_edge_phase: phase must be 1 or 2");
    return proc_edge_range(h, k, phase);
} MGN_CATCH(h)

int mgn_edge_boundary_tiles(const mgn_handle* h, int32_t set, int32_t* boundary, int32_t* total) try {
    if (!h || !h->have_graph) return MGN_E_STATE;
    if (set < 0 || set >= h->nsets) return MGN_E_ARG;
    if (boundary) *boundary = boundary_tiles(h, set);
    if (total) *total = h->es[set].ntiles_e;
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_proc_node(mgn_handle* h, int32_t k, int32_t project_next) try {
    if (int rc = need(h, true, true)) return rc;
    if (k < 0 || k >= h->cfg.mps) return fail(h, MGN_E_ARG, "mgn_proc_node: step %d out of range", k);
    if (project_next && k + 1 >= h->cfg.mps) return fail(h, MGN_E_ARG, "mgn_proc_node: no step %d to project for", k + 1);
    ProfScope ps(h, F_NODE);
    if (is_bf16(h) && !use_c16(h)) {
        HIPCHK(h, launch_node_bf16(bf_node_args(h, k), h->stream));
        for (int q = 0; project_next && q < h->nsets; ++q)
            if (int rc = project_set(h, k, q)) return rc;
        return MGN_OK;
    }
    // large meshes: MLP and projection as two launches (the projection then has both of its chunks LDS-resident);
    // small meshes are launch-latency-bound: one fused launch (M-cyl: 53.0 -> 50.8 us per step)
    // (... and from two tiles per CU where the split-path node kernels exist: they are two launches by construction)
    const bool split_pair = !is_bf16(h) && h->cfg.L == 128 && k < (int)h->spoff.size() && h->spoff[k].have_n && h->wsp.p && node_split_size(h->ntiles_n);
    if (project_next && !(h->nsets == 2 && use_c16(h)) && (h->nsets > 1 || split_pair || (h->node_split && !launch_is_small(h->ntiles_n)))) {
        // MLP (2 of its chunks LDS-resident, the others stream from L2), then per edge set the projection with both of
        // its chunks resident -- or, one edge set on two fp16 pieces, both in one lock-step launch (k_node_ring_hs)
        if (h->nsets == 1 && split_pair) {
            bool fused = false;
            HIPCHK(h, launch_node_project_fused(h->cfg.L, node_args(h, k, 0), h->stream, &fused));
            if (fused) return MGN_OK;
        }
        HIPCHK(h, launch_node_step(h->cfg.L, node_args(h, k, 0), h->stream));
        for (int q = 0; q < h->nsets; ++q)
            if (int rc = project_set(h, k, q)) return rc;
        return MGN_OK;
    }
    const NodeArgs a = node_args(h, k, project_next ? 1 : 0);
    HIPCHK(h, launch_node_step(h->cfg.L, a, h->stream));
    return MGN_OK;
} MGN_CATCH(h)

int mgn_proc_node_phase(mgn_handle* h, int32_t k, int32_t phase) try {
    if (int rc = need(h, true, true)) return rc;
    if (k < -1 || k + 1 >= h->cfg.mps) return fail(h, MGN_E_ARG, "mgn_proc_node_phase: no step %d to project for", k + 1);
    if (phase != 1 && phase != 2) return fail(h, MGN_E_ARG, "mgn_proc_node_phase: phase must be 1 or 2");
    ProfScope ps(h, F_NODE);
    const int ntb_all = (h->g.n_boundary + TILE - 1) / TILE;   // tiles that contain a boundary node
    const int ntb = ntb_all < h->ntiles_n ? ntb_all : h->ntiles_n;
    if (phase == 1 && k >= 0) {
        if (is_bf16(h) && !use_c16(h)) HIPCHK(h, launch_node_bf16(bf_node_args(h, k), h->stream));
        else HIPCHK(h, launch_node_step(h->cfg.L, node_args(h, k, 0), h->stream));
    }
    const int32_t tile0 = phase == 1 ? 0 : ntb, nt = phase == 1 ? ntb : h->ntiles_n - ntb;
    for (int q = 0; q < h->nsets; ++q)
        if (int rc = project_set(h, k >= 0 ? k : h->cfg.mps, q, tile0, nt)) return rc;
    return MGN_OK;
} MGN_CATCH(h)

static int decode_impl(mgn_handle* h, bool use_norms) {
    const mgn_config& c = h->cfg;
    ProfScope ps(h, F_DEC);
    if (is_bf16(h)) HIPCHK(h, launch_tile_bf16_to_f32(h->bV.as<uint16_t>(), h->V.as<float>(), h->ntiles_n, h->stream));
    DecArgs a{};
    a.n = h->g.n_own;
    a.ntiles = h->ntiles_n;
    a.V = h->V.as<float>();
    a.w3f = W(h, h->de_w3f);
    a.b3 = W(h, h->de_b3);
    a.O = c.O;
    if (use_norms && h->have_onorm) {
        const float* nrm = h->norms.as<float>() + 2 * c.Fn + 2 * c.Fe;
        a.oscale = nrm;
        a.oshift = nrm + c.O;
    }
    a.mask = (use_norms && h->have_mask) ? h->d_mask.as<float>() : nullptr;
    a.gid = h->d_own_gid.as<int32_t>();
    a.out = h->out_override ? h->out_override : h->d_out.as<float>();
    for (int i = 0; i < 2; ++i) a.chunk[i] = W(h, h->de_ch[i]);
    for (int i = 0; i < 2; ++i) {
        a.splith[i] = (h->have_ench && h->wsp.p) ? h->wsp.as<uint16_t>() + h->deh_ch[i] : nullptr;
        a.h2_rs[i] = h->have_ench ? 1.f / h->deh_s[i] : 1.f;
    }
    a.tabs = W(h, h->de_tabs);
    a.gen = gen_of(h, h->de_gen, false);
    HIPCHK(h, launch_decode(c.L, a, h->stream));
    return MGN_OK;
}

int mgn_fwd_decode(mgn_handle* h) try {
    if (int rc = need(h, true, true)) return rc;
    return decode_impl(h, false);
} MGN_CATCH(h)

int mgn_fwd_download(mgn_handle* h, float* out) try {
    if (int rc = need(h, false, true)) return rc;
    if (!out) return fail(h, MGN_E_ARG, "mgn_fwd_download: null out");
    const LocalGraph& g = h->g;
    const int O = h->cfg.O;
    std::vector<float> loc((size_t)g.n_own * O);
    HIPCHK(h, hipMemcpyAsync(loc.data(), h->d_out.p, loc.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int32_t i = 0; i < g.n_own; ++i) memcpy(out + (size_t)g.own_gid[i] * O, loc.data() + (size_t)i * O, (size_t)O * 4);
    return MGN_OK;
} MGN_CATCH(h)

static int forward_partitioned(mgn_handle* h, const float* nf, const float* ef, float* out);
static int gather_rows_global(mgn_handle* h, const float* local_dev, int W, float* out);
static int processor_pass_staged(mgn_handle* h, int32_t nsteps, bool begin);
static int need_comm(mgn_handle* h, const char* who);

// the O x N state of a right-hand side: one partition takes it as it is, a partitioned handle the rows it owns
static int upload_state(mgn_handle* h, const float* x) {
    const LocalGraph& g = h->g;
    const int O = h->cfg.O;
    if (h->cfg.nranks == 1 && !g.renumbered) {
        HIPCHK(h, hipMemcpyAsync(h->d_nfA.p, x, (size_t)g.N * O * 4, hipMemcpyHostToDevice, h->stream));
        return MGN_OK;
    }
    std::vector<float> loc((size_t)g.n_own * O);
    for (int32_t i = 0; i < g.n_own; ++i) memcpy(loc.data() + (size_t)i * O, x + (size_t)g.own_gid[i] * O, (size_t)O * 4);
    HIPCHK(h, hipMemcpyAsync(h->d_nfA.p, loc.data(), loc.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MGN_OK;
}

static int run_processor(mgn_handle* h, int nsteps) {
    // partitioned: the staged schedule; P, Q of step 0 came from the encoder, so the pass starts with their exchange
    if (h->cfg.nranks != 1) return processor_pass_staged(h, nsteps, false);
    for (int k = 0; k < nsteps; ++k) {
        if (int rc = mgn_proc_edge(h, k)) return rc;
        if (int rc = mgn_proc_node(h, k, k + 1 < nsteps ? 1 : 0)) return rc;
    }
    return MGN_OK;
}

int mgn_forward(mgn_handle* h, const float* nf, const float* ef, float* out) try {
    if (h && !h->host_only && h->cfg.ln_dims == MGN_LN_ALL) return lnall_forward(h, nf, ef, out);
    if (int rc = need(h, true, true)) return rc;
    if (h->cfg.nranks != 1) return forward_partitioned(h, nf, ef, out);
    if (int rc = need_set_features(h)) return rc;
    const float* nfA_before = h->d_nfA.as<float>();
    const float* ef_before = h->es[0].d_ef.as<float>();
    const int wa_before = h->in_wa, wb_before = h->in_wb;
    if (int rc = mgn_fwd_upload(h, nf, ef)) return rc;
    if (h->fwd_exec && (nfA_before != h->d_nfA.as<float>() || ef_before != h->es[0].d_ef.as<float>() || wa_before != h->in_wa || wb_before != h->in_wb)) {
        (void)hipGraphExecDestroy(h->fwd_exec);   // an input buffer moved or changed its layout since the capture
        h->fwd_exec = nullptr;
        h->fwd_warm = false;
    }
    auto launches = [&]() -> int {
        if (int rc = encode_impl(h, false)) return rc;
        if (int rc = run_processor(h, h->cfg.mps)) return rc;
        return decode_impl(h, false);
    };
    if (int rc = run_graphed(h, h->fwd_exec, h->fwd_warm, launches)) return rc;
    return mgn_fwd_download(h, out);
} MGN_CATCH(h)

int mgn_set_static(mgn_handle* h, const float* onehot, const float* ef_raw, const float* val_mask) try {
    const bool lnall = h && h->cfg.ln_dims == MGN_LN_ALL;
    if (int rc = need(h, true, true, !lnall, true)) return rc;
    const mgn_config& c = h->cfg;
    if (c.nranks != 1) if (int rc = need_comm(h, "mgn_set_static")) return rc;
    if (h->nsets != 1) return fail(h, MGN_E_STATE, "mgn_set_static / mgn_ode_step / mgn_rollout mirror the reference's single-edge-set RHS (src/solve.jl:188-219)");
    if (!ef_raw || (c.Fn > c.O && !onehot)) return fail(h, MGN_E_ARG, "mgn_set_static: null argument");
    if (c.Fn < c.O) return fail(h, MGN_E_ARG, "mgn_set_static: Fn < O");
    const LocalGraph& g = h->g;
    invalidate_static(h);
    {   // static inputs through the common upload (a partitioned handle keeps the rows it owns); the state slot is a placeholder
        std::vector<float> x0((size_t)g.N * c.O, 0.f);
        if (int rc = upload_inputs(h, x0.data(), c.O, onehot, c.Fn - c.O, ef_raw, true)) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    h->have_mask = val_mask != nullptr;
    if (val_mask) {
        HIPCHK(h, h->d_mask.ensure((size_t)g.N * 4));
        HIPCHK(h, hipMemcpyAsync(h->d_mask.p, val_mask, (size_t)g.N * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (lnall) {   // whole-array LayerNorm: the unfused right-hand side (mgn_train.cpp) encodes the edges on its first evaluation
        if (int rc = lnall_rhs_prepare(h)) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->have_static = true;
        return MGN_OK;
    }
    if (int rc = encode_impl(h, true, false, true)) return rc;       // edge encoder: once per trajectory
    const bool bf = is_bf16(h);
    const size_t eb = tile_floats(h->es[0].ntiles_e, c.L) * (bf ? 2 : 4);
    HIPCHK(h, h->es[0].elat0.ensure(eb));
    HIPCHK(h, hipMemcpyAsync(h->es[0].elat0.p, bf ? h->es[0].bElat.p : h->es[0].Elat.p, eb, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_static = true;
    return MGN_OK;
} MGN_CATCH(h)

// ode_step under ln_dims = MGN_LN_ALL: build_graph's normalisation on the host, the unfused whole-array forward, inverse_data and
// val_mask on the host (reference src/graph.jl:80-93, src/solve.jl:198-218).  The one-shot form only; a hedge, not a fast path.
static int lnall_ode_step(mgn_handle* h, const float* x, const float* onehot, const float* ef_raw, const float* val_mask, float* dxdt) {
    const mgn_config& c = h->cfg;
    if (!h->have_params) return fail(h, MGN_E_STATE, "mgn_set_params has not been called");
    if (!h->have_graph) return fail(h, MGN_E_STATE, "mgn_set_graph has not been called");
    if (!x || !dxdt) return fail(h, MGN_E_ARG, "mgn_ode_step: null argument");
    if (c.Fn < c.O) return fail(h, MGN_E_ARG, "mgn_ode_step: Fn < O");
    if (!onehot && !ef_raw && !val_mask) {   // resident form: static inputs from mgn_set_static, only the state moves
        if (!h->have_static) return fail(h, MGN_E_STATE, "mgn_ode_step without static inputs: call mgn_set_static first or pass them");
        if (int rc = upload_state(h, x)) return rc;
        if (int rc = lnall_rhs_prepare(h)) return rc;
        HIPCHK(h, h->d_out.ensure((size_t)(h->g.n_own > 0 ? h->g.n_own : 1) * c.O * 4));
        if (!h->lnall_edges) {        // first evaluation of the trajectory: the edge encoder runs too (eager)
            if (int rc = lnall_rhs_dev(h, h->d_nfA.as<float>(), h->d_out.as<float>(), false)) return rc;
            h->lnall_edges = true;
        } else {                      // afterwards the ~130 launches of a right-hand side replay from one launch graph (small meshes)
            auto launches = [&]() -> int { return lnall_rhs_dev(h, h->d_nfA.as<float>(), h->d_out.as<float>(), true); };
            if (int rc = run_graphed(h, h->rhs_exec, h->rhs_warm, launches)) return rc;
        }
        return mgn_fwd_download(h, dxdt);
    }
    if (!ef_raw || (c.Fn > c.O && !onehot)) return fail(h, MGN_E_ARG, "mgn_ode_step: null argument");
    invalidate_static(h);   // the one-shot path overwrites the resident inputs
    const int64_t N = h->g.N, E = h->g.set[0].E;
    const float* nrm = h->norms_host.empty() ? nullptr : h->norms_host.data();   // [node scale, shift | edge scale, shift | out scale, shift]
    const float* ns = (nrm && h->have_nnorm) ? nrm : nullptr;
    const float* es = (nrm && h->have_enorm) ? nrm + 2 * c.Fn : nullptr;
    const float* os = (nrm && h->have_onorm) ? nrm + 2 * c.Fn + 2 * c.Fe : nullptr;
    std::vector<float> nf((size_t)N * c.Fn), ef((size_t)E * c.Fe), out((size_t)N * c.O);
    const int W1 = c.Fn - c.O;
    for (int64_t n = 0; n < N; ++n)
        for (int f = 0; f < c.Fn; ++f) {
            const float v = f < c.O ? x[n * c.O + f] : onehot[n * W1 + (f - c.O)];
            nf[(size_t)n * c.Fn + f] = ns ? v * ns[f] + ns[c.Fn + f] : v;
        }
    for (int64_t j = 0; j < E; ++j)
        for (int f = 0; f < c.Fe; ++f) ef[(size_t)j * c.Fe + f] = es ? ef_raw[j * c.Fe + f] * es[f] + es[c.Fe + f] : ef_raw[j * c.Fe + f];
    if (int rc = lnall_forward(h, nf.data(), ef.data(), out.data())) return rc;
    for (int64_t n = 0; n < N; ++n)
        for (int o = 0; o < c.O; ++o) {
            const float y = os ? out[(size_t)n * c.O + o] * os[o] + os[c.O + o] : out[(size_t)n * c.O + o];
            dxdt[n * c.O + o] = val_mask ? y * val_mask[n] : y;
        }
    return MGN_OK;
}

int mgn_ode_step(mgn_handle* h, const float* x, const float* onehot, const float* ef_raw, const float* val_mask, float* dxdt) try {
    if (h && !h->host_only && h->cfg.ln_dims == MGN_LN_ALL) return lnall_ode_step(h, x, onehot, ef_raw, val_mask, dxdt);
    if (int rc = need(h, true, true)) return rc;
    const mgn_config& c = h->cfg;
    if (h->cfg.nranks != 1) if (int rc = need_comm(h, "mgn_ode_step")) return rc;
    if (h->nsets != 1) return fail(h, MGN_E_STATE, "%s mirrors the reference's single-edge-set RHS (src/solve.jl:188-219); this handle has two edge sets", "mgn_ode_step");
    if (!x || !dxdt) return fail(h, MGN_E_ARG, "mgn_ode_step: null argument");
    if (c.Fn < c.O) return fail(h, MGN_E_ARG, "mgn_ode_step: Fn < O");
    if (!onehot && !ef_raw && !val_mask) {
        // fast path: static inputs and encoded edges are resident (mgn_set_static); only the state moves
        if (!h->have_static) return fail(h, MGN_E_STATE, "mgn_ode_step without static inputs: call mgn_set_static first or pass them");
        if (int rc = upload_state(h, x)) return rc;
        auto launches = [&]() -> int {
            if (int rc = encode_impl(h, true, true, false)) return rc;
            const bool bf = is_bf16(h);
            const size_t eb = tile_floats(h->es[0].ntiles_e, c.L) * (bf ? 2 : 4);
            if (elat_src_ok(h)) h->elat_src_override = h->es[0].elat0.as<float>();
            else HIPCHK(h, hipMemcpyAsync(bf ? h->es[0].bElat.p : h->es[0].Elat.p, h->es[0].elat0.p, eb, hipMemcpyDeviceToDevice, h->stream));
            const int rcp = run_processor(h, c.mps);
            h->elat_src_override = nullptr;
            if (rcp) return rcp;
            return decode_impl(h, true);
        };
        // a Julia-driven solve calls this once per right-hand side
        if (c.nranks != 1) {        // (no hipGraph around communicator calls)
            if (int rc = launches()) return rc;
            return gather_rows_global(h, h->d_out.as<float>(), c.O, dxdt);
        }
        if (int rc = run_graphed(h, h->rhs_exec, h->rhs_warm, launches)) return rc;
        return mgn_fwd_download(h, dxdt);
    }
    if (!ef_raw || (c.Fn > c.O && !onehot)) return fail(h, MGN_E_ARG, "mgn_ode_step: null argument");
    invalidate_static(h);   // the one-shot path overwrites the resident inputs
    if (int rc = upload_inputs(h, x, c.O, onehot, c.Fn - c.O, ef_raw)) return rc;
    h->have_mask = val_mask != nullptr;
    if (val_mask) {
        HIPCHK(h, h->d_mask.ensure((size_t)h->g.N * 4));
        HIPCHK(h, hipMemcpyAsync(h->d_mask.p, val_mask, (size_t)h->g.N * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (int rc = encode_impl(h, true)) return rc;
    if (int rc = run_processor(h, c.mps)) return rc;
    if (int rc = decode_impl(h, true)) return rc;
    if (c.nranks != 1) return gather_rows_global(h, h->d_out.as<float>(), c.O, dxdt);    // the complete dx/dt on every rank
    return mgn_fwd_download(h, dxdt);
} MGN_CATCH(h)

// ---- native rollout driver (N1) -----------------------------------------------------------------------------
namespace {

// Tsitouras 5(4) tableau (the method OrdinaryDiffEq.jl calls Tsit5)
const double TS_C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
const double TS_A[7][6] = {
    {0, 0, 0, 0, 0, 0},
    {0.161, 0, 0, 0, 0, 0},
    {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
    {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0},
    {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};
const double TS_BT[7] = {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
                         0.5823571654525552, -0.45808210592918697, 0.015151515151515152};

struct Rollout {
    mgn_engine* h;
    mgn_rollout_desc* d;
    int64_t n;                 // rows * O of the state this handle integrates (all N rows, or the owned rows of a partition)
    int64_t n_global = 0;      // N * O
    int32_t nrows = 0;
    float *u, *unew, *utmp, *k[7], *frames, *saves;
    uint8_t* mask;
    double* partial;
    int n_rhs = 0;
    // the solver's time type (mgn_rollout_desc.time_f64): Float32 times are held in doubles and rounded after every operation
    // (a double operation on two floats, rounded to float, IS the float operation)
    bool f64 = false;
    double sdt = 0.0;          // saves_dt in that type
    double tt(double v) const { return f64 ? v : (double)(float)v; }

    // One right-hand side is ~35 launches; on a small mesh they are latency-bound, so each distinct (x, kout) pair of the
    // solver (1 for Euler, 7 for Tsit5) gets its launch sequence captured once and replayed (hipGraph).
    struct RhsGraph { float* x; float* kout; hipGraphExec_t exec; };
    std::vector<RhsGraph> graphs;
    bool warmed = false;

    bool lnall_edges_done = false;
    int rhs_launches(float* x, float* kout) {
        const mgn_config& c = h->cfg;
        if (c.ln_dims == MGN_LN_ALL) {     // the unfused whole-array right-hand side (mgn_train.cpp); its first evaluation encodes the edges
            const int rc = lnall_rhs_dev(h, x, kout, lnall_edges_done);
            lnall_edges_done = true;
            return rc;
        }
        h->srcA_override = x;
        h->out_override = kout;
        int rc = encode_impl(h, true, true, false);
        if (!rc) {
            // encoded edge latents are identical for every RHS of a trajectory (static edge features, frozen e_norm)
            const bool bf = c.dtype == MGN_BF16;
            const size_t eb = tile_floats(h->es[0].ntiles_e, c.L) * (bf ? 2 : 4);
            if (elat_src_ok(h)) {
                h->elat_src_override = reinterpret_cast<const float*>(h->ode.as<char>() + elat0_off);
            } else {
                hipError_t e = hipMemcpyAsync(bf ? h->es[0].bElat.p : h->es[0].Elat.p, h->ode.as<char>() + elat0_off, eb, hipMemcpyDeviceToDevice, h->stream);
                if (e != hipSuccess) rc = fail(h, MGN_E_HIP, "rollout: Elat restore failed: %s", hipGetErrorString(e));
            }
        }
        if (!rc) rc = run_processor(h, c.mps);
        h->elat_src_override = nullptr;
        if (!rc) rc = decode_impl(h, true);
        h->srcA_override = nullptr;
        h->out_override = nullptr;
        return rc;
    }

    // f(x, t): in-place inflow overwrite of x, then dx/dt -> kout    (ode_func_eval, reference src/solve.jl:147-158)
    int rhs(float* x, double t, float* kout) {
        const mgn_config& c = h->cfg;
        if (mask && frames) {
            // data[field][:, :, floor(Int, t / saves_dt) + 1] (reference src/solve.jl:151): the quotient in the solver's own time type,
            // no tolerance -- a t that sits an ulp below a frame boundary re-uses the previous frame there too -- and an index outside
            // the data is the reference's BoundsError.  MGN_INFLOW_TOLERANT: nearest-below with a guard of 1e-3 frames (a Float32 time drifts by ~1e-4 frames), clamped.
            int64_t fr;
            if (d->inflow_rule == MGN_INFLOW_TOLERANT) {
                fr = (int64_t)std::floor(t / sdt + 1e-3);
                if (fr < 0) fr = 0;
                if (fr >= d->n_frames) fr = d->n_frames - 1;
            } else {
                fr = (int64_t)std::floor(tt(t / sdt));
                if (fr < 0 || fr >= d->n_frames)
                    return fail(h, MGN_E_ARG, "mgn_rollout: inflow frame %lld at t = %.9g is outside the %d frames given (reference: BoundsError)",
                                (long long)fr, t, d->n_frames);
            }
            HIPCHK(h, launch_overwrite(x, frames + (size_t)fr * n, mask, nrows, c.O, h->stream));
        }
        ++n_rhs;
        const bool graphable = h->use_graph && !h->prof && h->stream != nullptr && h->cfg.nranks == 1 && launch_is_small(h->ntiles_n);
        if (!graphable || !warmed) {       // the first RHS runs eagerly: it sets the per-kernel attributes outside of any capture
            warmed = true;
            return rhs_launches(x, kout);
        }
        for (const RhsGraph& g : graphs)
            if (g.x == x && g.kout == kout) {
                HIPCHK(h, hipGraphLaunch(g.exec, h->stream));
                return MGN_OK;
            }
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            h->use_graph = 0;
            return rhs_launches(x, kout);
        }
        const int rc = rhs_launches(x, kout);
        const hipError_t ce = hipStreamEndCapture(h->stream, &graph);
        hipGraphExec_t exec = nullptr;
        if (rc != MGN_OK || ce != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            h->use_graph = 0;              // eager from here on
            if (rc != MGN_OK) return rc;
            return rhs_launches(x, kout);
        }
        (void)hipGraphDestroy(graph);
        graphs.push_back({x, kout, exec});
        HIPCHK(h, hipGraphLaunch(exec, h->stream));
        return MGN_OK;
    }
    ~Rollout() {
        for (RhsGraph& g : graphs) (void)hipGraphExecDestroy(g.exec);
    }
    size_t elat0_off = 0;

    int norm(const float* a, const float* b, const LinComb& lc, float dt, double* out) {
        const int np_ = errnorm_partials();
        HIPCHK(h, launch_errnorm(a, b, lc, dt, d->abstol, d->reltol, n, partial, h->stream));
        std::vector<double> r(np_);
        HIPCHK(h, hipMemcpyAsync(r.data(), partial, np_ * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        double s = 0;
        for (double v : r) s += v;
        if (h->cfg.nranks != 1) {      // the same bits on every rank -> the same accept / reject decisions
            if (h->comm->allreduce_f64(&s, 1, 0, h->stream) != 0) return fail(h, MGN_E_RCCL, "rollout: error-norm reduction failed: %s", h->comm->err.c_str());
        }
        const int64_t ng = n_global > 0 ? n_global : n;
        *out = std::sqrt(s / (double)(ng > 0 ? ng : 1));
        return MGN_OK;
    }
};

}  // namespace

int mgn_rollout(mgn_handle* h, mgn_rollout_desc* d) try {
    const bool lnall = h && h->cfg.ln_dims == MGN_LN_ALL;
    if (int rc = need(h, true, true, !lnall, true)) return rc;
    const mgn_config& c = h->cfg;
    const bool part = c.nranks != 1;      // partitioned: every rank integrates the rows it owns; error norms are reduced over the ranks
    if (part) if (int rc = need_comm(h, "mgn_rollout")) return rc;
    if (h->nsets != 1) return fail(h, MGN_E_STATE, "%s mirrors the reference's single-edge-set RHS (src/solve.jl:188-219); this handle has two edge sets", "mgn_rollout");
    if (!d || !d->x0 || !d->out || !d->ef_raw || (c.Fn > c.O && !d->node_type_onehot)) return fail(h, MGN_E_ARG, "mgn_rollout: null argument");
    if (c.Fn < c.O) return fail(h, MGN_E_ARG, "mgn_rollout: Fn < O");
    const bool f64 = d->time_f64 != 0;
    const double T0 = f64 ? d->t0_f64 : (double)d->t0, T1 = f64 ? d->t1_f64 : (double)d->t1, DT = f64 ? d->dt_f64 : (double)d->dt,
                 SDT = f64 ? d->saves_dt_f64 : (double)d->saves_dt;
    if (d->n_saves < 1 || !(SDT > 0.0) || T1 < T0) return fail(h, MGN_E_ARG, "mgn_rollout: bad time grid");
    if (d->solver == 0 && !(DT > 0.0)) return fail(h, MGN_E_ARG, "mgn_rollout: Euler needs dt > 0");
    if (d->inflow_rule != MGN_INFLOW_REFERENCE && d->inflow_rule != MGN_INFLOW_TOLERANT) return fail(h, MGN_E_ARG, "mgn_rollout: unknown inflow_rule");
    if (d->solver != 0 && d->solver != 1) return fail(h, MGN_E_ARG, "mgn_rollout: solver must be 0 (Euler) or 1 (Tsit5)");
    if ((d->inflow_mask != nullptr) != (d->inflow_data != nullptr)) return fail(h, MGN_E_ARG, "mgn_rollout: inflow mask and data go together");
    if (d->solver == 1 && (d->abstol <= 0.f || d->reltol <= 0.f)) return fail(h, MGN_E_ARG, "mgn_rollout: tolerances must be > 0");
    const LocalGraph& g = h->g;
    invalidate_static(h);
    Rollout R;
    R.h = h;
    R.d = d;
    R.f64 = f64;
    R.sdt = SDT;
    auto tt = [&](double v) { return R.tt(v); };
    const bool loc = part || g.renumbered;            // the state lives in the engine's node order (owned rows): gathered in, scattered out
    const int32_t nloc = part ? g.n_own : g.N;        // rows of the state this handle integrates
    R.n = (int64_t)nloc * c.O;
    R.n_global = (int64_t)g.N * c.O;
    R.nrows = nloc;
    const size_t nb = (size_t)R.n * 4;
    const size_t fb = d->inflow_data ? (size_t)d->n_frames * nb : 0, sb = (size_t)d->n_saves * nb;
    const size_t eb = tile_floats(h->es[0].ntiles_e, c.L) * 4;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_u = take(nb), o_un = take(nb), o_ut = take(nb);
    size_t o_k[7];
    for (auto& o : o_k) o = take(nb);
    const size_t o_fr = take(fb), o_sv = take(sb), o_mask = take((size_t)nloc), o_part = take(errnorm_partials() * sizeof(double));
    R.elat0_off = take(eb);
    HIPCHK(h, h->ode.ensure(off));
    char* base = h->ode.as<char>();
    R.u = (float*)(base + o_u); R.unew = (float*)(base + o_un); R.utmp = (float*)(base + o_ut);
    for (int j = 0; j < 7; ++j) R.k[j] = (float*)(base + o_k[j]);
    R.frames = d->inflow_data ? (float*)(base + o_fr) : nullptr;
    R.saves = (float*)(base + o_sv);
    R.mask = d->inflow_mask ? (uint8_t*)(base + o_mask) : nullptr;
    R.partial = (double*)(base + o_part);

    if (!loc) {
        HIPCHK(h, hipMemcpyAsync(R.u, d->x0, nb, hipMemcpyHostToDevice, h->stream));
        if (R.frames) HIPCHK(h, hipMemcpyAsync(R.frames, d->inflow_data, fb, hipMemcpyHostToDevice, h->stream));
        if (R.mask) HIPCHK(h, hipMemcpyAsync(R.mask, d->inflow_mask, (size_t)g.N, hipMemcpyHostToDevice, h->stream));
    } else {        // the owned rows of the state, of every inflow frame and of the inflow mask
        const int O = c.O;
        std::vector<float> loc((size_t)nloc * O * (1 + (d->inflow_data ? d->n_frames : 0)));
        std::vector<uint8_t> lm(d->inflow_mask ? (size_t)nloc : 0);
        for (int32_t i = 0; i < nloc; ++i) {
            const size_t gi = (size_t)g.own_gid[i];
            memcpy(loc.data() + (size_t)i * O, d->x0 + gi * O, (size_t)O * 4);
            for (int f = 0; d->inflow_data && f < d->n_frames; ++f)
                memcpy(loc.data() + ((size_t)(1 + f) * nloc + i) * O, d->inflow_data + ((size_t)f * g.N + gi) * O, (size_t)O * 4);
            if (d->inflow_mask) lm[i] = d->inflow_mask[gi];
        }
        HIPCHK(h, hipMemcpyAsync(R.u, loc.data(), nb, hipMemcpyHostToDevice, h->stream));
        if (R.frames) HIPCHK(h, hipMemcpyAsync(R.frames, loc.data() + (size_t)nloc * O, fb, hipMemcpyHostToDevice, h->stream));
        if (R.mask) HIPCHK(h, hipMemcpyAsync(R.mask, lm.data(), (size_t)nloc, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    // static inputs: one-hot node types, raw edge features, val_mask; the edge encoder runs ONCE per trajectory
    if (int rc = upload_inputs(h, d->x0, c.O, d->node_type_onehot, c.Fn - c.O, d->ef_raw, true)) return rc;
    h->have_mask = d->val_mask != nullptr;
    if (d->val_mask) {
        HIPCHK(h, h->d_mask.ensure((size_t)g.N * 4));
        HIPCHK(h, hipMemcpyAsync(h->d_mask.p, d->val_mask, (size_t)g.N * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (lnall) {
        if (int rc = lnall_rhs_prepare(h)) return rc;
    } else {
        if (int rc = encode_impl(h, true, false, true)) return rc;
        HIPCHK(h, hipMemcpyAsync(base + R.elat0_off, is_bf16(h) ? h->es[0].bElat.p : h->es[0].Elat.p, is_bf16(h) ? eb / 2 : eb, hipMemcpyDeviceToDevice, h->stream));
    }

    d->n_accept = d->n_reject = 0;
    int saved = 0;
    auto save = [&]() -> hipError_t {
        if (saved >= d->n_saves) return hipSuccess;
        return hipMemcpyAsync(R.saves + (size_t)saved++ * R.n, R.u, nb, hipMemcpyDeviceToDevice, h->stream);
    };
    auto stop_time = [&](int i) { return tt(T0 + (double)i * SDT); };
    double t = T0;
    HIPCHK(h, save());   // solution at t0

    if (d->solver == 0) {
        const double dt = DT;
        const int64_t nsteps = (int64_t)std::llround((T1 - T0) / dt);
        for (int64_t i = 0; i < nsteps; ++i) {
            if (int rc = R.rhs(R.u, t, R.k[0])) return rc;
            LinComb lc{1, {1.f}, {R.k[0]}};
            HIPCHK(h, launch_lincomb(R.u, R.u, lc, (float)dt, R.n, h->stream));
            // the time the right-hand side sees is the integrator's own: t <- t + dt in its time type, step after step (a fixed-step
            // solve has no stops to snap to but the end of the interval); t0 + (i + 1) dt would floor differently at frame boundaries
            t = (i + 1 == nsteps && std::fabs(tt(t + dt) - T1) <= 1e-5 * SDT) ? T1 : tt(t + dt);
            ++d->n_accept;
            // saveat: the state of the step that ends at the save point.  In its own time type the integrator's t drifts off the
            // save grid by a few ulps per step (Float32: ~1e-6 s after 600 steps of 0.01 s); the reference interpolates there, which
            // moves the saved state by (drift / dt) of one step's change -- far below the rollout tolerance -- so: the nearest step.
            while (saved < d->n_saves && stop_time(saved) <= t + 0.25 * dt) HIPCHK(h, save());
        }
    } else {
        // adaptive Tsit5, PI controller (beta1 = 7/50, beta2 = 2/25, gamma = 0.9, qmin = 0.2, qmax = 10), tstops = saves
        const double beta1 = 7.0 / 50, beta2 = 2.0 / 25, gamma = 0.9, qmin = 0.2, qmax = 10.0;
        double qold = 1e-4;
        if (int rc = R.rhs(R.u, t, R.k[0])) return rc;     // k1 (FSAL afterwards)
        double dt = DT;
        if (dt <= 0) {   // Hairer-Wanner starting step
            LinComb z{0, {}, {}};
            double d0, d1, d2;
            LinComb l1{1, {1.f}, {R.k[0]}};
            // d0 = ||u||, d1 = ||f0|| in the scaled norm: errnorm(dt = 1) of u and k1 themselves
            LinComb lu{1, {1.f}, {R.u}};
            if (int rc = R.norm(R.u, R.u, lu, 1.f, &d0)) return rc;
            if (int rc = R.norm(R.u, R.u, l1, 1.f, &d1)) return rc;
            const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
            HIPCHK(h, launch_lincomb(R.utmp, R.u, l1, (float)h0, R.n, h->stream));
            if (int rc = R.rhs(R.utmp, tt(t + h0), R.k[1])) return rc;
            LinComb ld{2, {1.f, -1.f}, {R.k[1], R.k[0]}};
            if (int rc = R.norm(R.u, R.u, ld, (float)(1.0 / h0), &d2)) return rc;
            const double mx = d1 > d2 ? d1 : d2;
            const double h1 = mx <= 1e-15 ? (h0 * 1e-3 > 1e-6 ? h0 * 1e-3 : 1e-6) : std::pow(0.01 / mx, 1.0 / 5);
            dt = 100 * h0 < h1 ? 100 * h0 : h1;
            (void)z;
        }
        const double tend = T1;
        int guard = 0;
        // float32 descriptors: t1 and n*saves_dt may differ in the last ulp; an interval shorter than 1e-5 save
        // periods is not worth a step
        while (t < tend - 1e-5 * SDT && ++guard < 10000000) {
            double tstop = saved < d->n_saves ? stop_time(saved) : tend;
            if (tstop > tend) tstop = tend;
            bool hit_stop = false;
            double hstep = dt;
            if (t + hstep >= tstop - 1e-9 * std::fabs(tstop)) { hstep = tstop - t; hit_stop = true; }
            for (int sidx = 1; sidx < 7; ++sidx) {     // stages 2..7; stage 7 is evaluated on unew (FSAL)
                LinComb lc{sidx, {}, {}};
                for (int j = 0; j < sidx; ++j) { lc.c[j] = (float)TS_A[sidx][j]; lc.k[j] = R.k[j]; }
                float* dst = (sidx == 6) ? R.unew : R.utmp;
                HIPCHK(h, launch_lincomb(dst, R.u, lc, (float)hstep, R.n, h->stream));
                // (a stage that lands on the stop itself sees the stop's time: c7 = 1)
                if (int rc = R.rhs(dst, (sidx == 6 && hit_stop) ? tstop : tt(t + tt(TS_C[sidx] * hstep)), R.k[sidx])) return rc;
            }
            LinComb le{7, {}, {}};
            for (int j = 0; j < 7; ++j) { le.c[j] = (float)TS_BT[j]; le.k[j] = R.k[j]; }
            double EEst;
            if (int rc = R.norm(R.u, R.unew, le, (float)hstep, &EEst)) return rc;
            if (!(EEst == EEst)) return fail(h, MGN_E_STATE, "mgn_rollout: NaN in the error estimate at t = %g", t);
            const double q11 = std::pow(EEst > 1e-30 ? EEst : 1e-30, beta1);
            if (EEst <= 1.0) {
                double q = q11 / std::pow(qold, beta2);
                q = std::max(1.0 / qmax, std::min(1.0 / qmin, q / gamma));
                qold = std::max(EEst, 1e-4);
                std::swap(R.u, R.unew);
                std::swap(R.k[0], R.k[6]);            // FSAL: k1 of the next step = f(unew)
                t = hit_stop ? tstop : tt(t + hstep);
                ++d->n_accept;
                if (!hit_stop || hstep >= dt * (1 - 1e-9)) dt = hstep / q;   // a step cut by a stop does not shrink dt
                else dt = std::max(dt, hstep / q);
                if (hit_stop && saved < d->n_saves && std::fabs(stop_time(saved) - t) <= 1e-9 * std::fabs(t) + 1e-12) HIPCHK(h, save());
            } else {
                ++d->n_reject;
                dt = hstep / std::min(1.0 / qmin, q11 / gamma);
            }
        }
    }
    while (saved < d->n_saves) HIPCHK(h, save());      // (t1 short of the last stop: repeat the final state)
    if (part) {     // every rank returns the complete solution
        for (int i = 0; i < d->n_saves; ++i)
            if (int rc = gather_rows_global(h, R.saves + (size_t)i * R.n, c.O, d->out + (size_t)i * g.N * c.O)) return rc;
    } else if (loc) {   // one partition in the engine's own node order: back to the caller's
        std::vector<float> sv((size_t)d->n_saves * R.n);
        HIPCHK(h, hipMemcpyAsync(sv.data(), R.saves, sb, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < d->n_saves; ++i)
            for (int32_t j = 0; j < g.n_own; ++j)
                memcpy(d->out + ((size_t)i * g.N + (size_t)g.own_gid[j]) * c.O, sv.data() + ((size_t)i * g.n_own + j) * c.O, (size_t)c.O * 4);
    } else {
        HIPCHK(h, hipMemcpyAsync(d->out, R.saves, sb, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    d->n_rhs = R.n_rhs;
    return MGN_OK;
} MGN_CATCH(h)

// ---- latents -------------------------------------------------------------------------------------
// Host boundary of the latents: the caller's arrays travel over PCIe as they are (one contiguous copy each) and
// the gather into engine order / tile-major storage runs on the device.
static int import_edges(mgn_handle* h, int q, const float* e) {
    const EdgeTopo& t = h->g.set[q];
    auto& es = h->es[q];
    const int L = h->cfg.L;
    if (t.e_local > 0) {
        if (!e) return fail(h, MGN_E_ARG, "latents import: null edge input");
        const size_t ebytes = (size_t)t.E * L * 4;
        HIPCHK(h, h->stage.ensure(ebytes));
        HIPCHK(h, hipMemcpyAsync(h->stage.p, e, ebytes, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, launch_rows_to_tiles(h->stage.as<float>(), es.d_edge_gid.as<int64_t>(), nullptr, es.Elat.as<float>(), t.e_local, L, h->stream));
    }
    // bf16 mode: the processor state proper is the bf16 copy (rounded once, on the device)
    if (is_bf16(h)) HIPCHK(h, launch_tile_f32_to_bf16(es.Elat.as<float>(), es.bElat.as<uint16_t>(), es.ntiles_e, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MGN_OK;
}

static int export_edges(mgn_handle* h, int q, float* e) {
    const EdgeTopo& t = h->g.set[q];
    auto& es = h->es[q];
    const int L = h->cfg.L;
    if (!e || t.E == 0) return MGN_OK;
    if (is_bf16(h)) HIPCHK(h, launch_tile_bf16_to_f32(es.bElat.as<uint16_t>(), es.Elat.as<float>(), es.ntiles_e, h->stream));
    const size_t ebytes = (size_t)t.E * L * 4;
    HIPCHK(h, h->stage.ensure(ebytes));
    // single partition: every row is owned, the staging image is complete; else keep the caller's foreign rows
    if (h->cfg.nranks != 1) HIPCHK(h, hipMemcpyAsync(h->stage.p, e, ebytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_tiles_to_rows(es.Elat.as<float>(), es.d_edge_gid.as<int64_t>(), nullptr, h->stage.as<float>(), t.e_local, L, h->stream));
    HIPCHK(h, hipMemcpyAsync(e, h->stage.p, ebytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MGN_OK;
}

int mgn_latents_import(mgn_handle* h, const float* v, const float* e) try {
    if (int rc = need(h, false, true)) return rc;
    if (!v) return fail(h, MGN_E_ARG, "mgn_latents_import: null input");
    const LocalGraph& g = h->g;
    const int L = h->cfg.L;
    const size_t vb = (size_t)g.N * L * 4;
    HIPCHK(h, h->stage.ensure(vb));
    HIPCHK(h, hipMemcpyAsync(h->stage.p, v, vb, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, launch_rows_to_tiles(h->stage.as<float>(), nullptr, h->d_own_gid.as<int32_t>(), h->V.as<float>(), g.n_own, L, h->stream));
    if (is_bf16(h)) HIPCHK(h, launch_tile_f32_to_bf16(h->V.as<float>(), h->bV.as<uint16_t>(), h->ntiles_n, h->stream));
    return import_edges(h, 0, e);
} MGN_CATCH(h)

// Writes the owned rows into the caller's GLOBAL-shaped arrays (other rows are left untouched).
int mgn_latents_export(mgn_handle* h, float* v, float* e) try {
    if (int rc = need(h, false, true)) return rc;
    const LocalGraph& g = h->g;
    const int L = h->cfg.L;
    if (v) {
        if (is_bf16(h)) HIPCHK(h, launch_tile_bf16_to_f32(h->bV.as<uint16_t>(), h->V.as<float>(), h->ntiles_n, h->stream));
        const size_t vb = (size_t)g.N * L * 4;
        HIPCHK(h, h->stage.ensure(vb));
        if (h->cfg.nranks != 1) HIPCHK(h, hipMemcpyAsync(h->stage.p, v, vb, hipMemcpyHostToDevice, h->stream));   // keep foreign rows
        HIPCHK(h, launch_tiles_to_rows(h->V.as<float>(), nullptr, h->d_own_gid.as<int32_t>(), h->stage.as<float>(), g.n_own, L, h->stream));
        HIPCHK(h, hipMemcpyAsync(v, h->stage.p, vb, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return export_edges(h, 0, e);
} MGN_CATCH(h)

int mgn_edge_latents_import(mgn_handle* h, int32_t set, const float* e) try {
    if (int rc = need(h, false, true)) return rc;
    if (set < 0 || set >= h->nsets) return fail(h, MGN_E_ARG, "mgn_edge_latents_import: set %d out of range", set);
    return import_edges(h, set, e);
} MGN_CATCH(h)

int mgn_edge_latents_export(mgn_handle* h, int32_t set, float* e) try {
    if (int rc = need(h, false, true)) return rc;
    if (set < 0 || set >= h->nsets) return fail(h, MGN_E_ARG, "mgn_edge_latents_export: set %d out of range", set);
    return export_edges(h, set, e);
} MGN_CATCH(h)

int mgn_latents_randn(mgn_handle* h, uint64_t seed) try {
    if (int rc = need(h, false, true)) return rc;
    const LocalGraph& g = h->g;
    HIPCHK(h, launch_randn_rows(h->V.as<float>(), nullptr, h->d_own_gid.as<int32_t>(), g.n_own, h->cfg.L, seed, h->stream));
    for (int q = 0; q < h->nsets; ++q)
        HIPCHK(h, launch_randn_rows(h->es[q].Elat.as<float>(), h->es[q].d_edge_gid.as<int64_t>(), nullptr, g.set[q].e_local, h->cfg.L,
                                    seed ^ (0xE5E5E5E5E5E5E5E5ull + (uint64_t)q * 0x9E3779B97F4A7C15ull), h->stream));
    if (is_bf16(h)) {
        HIPCHK(h, launch_tile_f32_to_bf16(h->V.as<float>(), h->bV.as<uint16_t>(), h->ntiles_n, h->stream));
        for (int q = 0; q < h->nsets; ++q)
            HIPCHK(h, launch_tile_f32_to_bf16(h->es[q].Elat.as<float>(), h->es[q].bElat.as<uint16_t>(), h->es[q].ntiles_e, h->stream));
    }
    return MGN_OK;
} MGN_CATCH(h)

// (sum, sum of squares) of the node latents and of the edge latents (all sets)
int mgn_latents_checksum(mgn_handle* h, double* sv, double* se, double* qv, double* qe) try {
    if (int rc = need(h, false, true)) return rc;
    const int np_ = checksum_partials();
    const int nb = 1 + h->nsets;
    if (is_bf16(h)) {
        HIPCHK(h, launch_tile_bf16_to_f32(h->bV.as<uint16_t>(), h->V.as<float>(), h->ntiles_n, h->stream));
        for (int q = 0; q < h->nsets; ++q)
            HIPCHK(h, launch_tile_bf16_to_f32(h->es[q].bElat.as<uint16_t>(), h->es[q].Elat.as<float>(), h->es[q].ntiles_e, h->stream));
    }
    HIPCHK(h, h->d_sum.ensure((size_t)nb * np_ * sizeof(double)));
    HIPCHK(h, launch_checksum(h->V.as<float>(), (int64_t)tile_floats(h->ntiles_n, h->cfg.L), h->d_sum.as<double>(), h->stream));
    for (int q = 0; q < h->nsets; ++q)
        HIPCHK(h, launch_checksum(h->es[q].Elat.as<float>(), (int64_t)tile_floats(h->es[q].ntiles_e, h->cfg.L),
                                  h->d_sum.as<double>() + (size_t)(1 + q) * np_, h->stream));
    std::vector<double> r((size_t)nb * np_);
    HIPCHK(h, hipMemcpyAsync(r.data(), h->d_sum.p, r.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double acc[4] = {0, 0, 0, 0};   // bitwise reproducible: fixed partials, fixed order
    for (int b = 0; b < np_ / 2; ++b) {
        acc[0] += r[2 * b];
        acc[1] += r[2 * b + 1];
    }
    for (int q = 0; q < h->nsets; ++q)
        for (int b = 0; b < np_ / 2; ++b) {
            acc[2] += r[(size_t)(1 + q) * np_ + 2 * b];
            acc[3] += r[(size_t)(1 + q) * np_ + 2 * b + 1];
        }
    if (sv) *sv = acc[0];
    if (qv) *qv = acc[1];
    if (se) *se = acc[2];
    if (qe) *qe = acc[3];
    return MGN_OK;
} MGN_CATCH(h)


// ---- communicator and the staged multi-partition schedule -------------------------------------------
#define COMMCHK(h, expr)                                                                                   \
    do {                                                                                                   \
        if ((expr) != 0) return fail(h, MGN_E_RCCL, "%s: %s", #expr, (h)->comm ? (h)->comm->err.c_str() : "no communicator"); \
    } while (0)

static int need_comm(mgn_handle* h, const char* who) {
    if (!h->comm) return fail(h, MGN_E_RCCL, "%s with nranks = %d needs a communicator: call mgn_comm_init on every rank first", who, h->cfg.nranks);
    return MGN_OK;
}

// per-peer byte layout of one exchange; rebuilt after every mgn_set_graph / mgn_set_edge_set
static int halo_plan(mgn_handle* h) {
    if (h->hx_ready) return MGN_OK;
    const LocalGraph& g = h->g;
    const int P = h->cfg.nranks;
    const size_t rowb = (size_t)mgn_halo_bytes_per_row(h);
    h->hx_sb.assign(P, 0); h->hx_so.assign(P, 0); h->hx_rb.assign(P, 0); h->hx_ro.assign(P, 0);
    size_t so = 0, ro = 0;
    for (int q = 0; q < P; ++q) {
        h->hx_sb[q] = (size_t)g.send_rows[q] * rowb; h->hx_so[q] = so; so += h->hx_sb[q];
        h->hx_rb[q] = (size_t)g.recv_rows[q] * rowb; h->hx_ro[q] = ro; ro += h->hx_rb[q];
    }
    // one edge set: the halo rows of P are one contiguous block in owner-rank order -- the layout of the receive buffer --
    // so the rows land there directly and no unpack copy runs
    h->hx_direct = h->nsets == 1 && !prows_blocked();   // (P rows are not contiguous: blocks of eight)
    if (!h->host_only) {
        HIPCHK(h, h->halo_send.ensure(so ? so : 16));
        if (!h->hx_direct) HIPCHK(h, h->halo_recv.ensure(ro ? ro : 16));
    }
    h->hx_ready = true;
    return MGN_OK;
}

static int exchange_start(mgn_handle* h) {
    if (int rc = halo_plan(h)) return rc;
    if (int rc = mgn_halo_pack(h, h->halo_send.p)) return rc;
    void* recv = h->halo_recv.p;
    if (h->hx_direct) {
        const size_t b = is_bf16(h) ? 2 : 4;
        recv = reinterpret_cast<char*>(is_bf16(h) ? h->es[0].bP.p : h->es[0].P.p) + (size_t)h->g.n_own * h->cfg.L * b;
    }
    COMMCHK(h, h->comm->a2a_start(h->halo_send.p, h->hx_sb.data(), h->hx_so.data(), recv, h->hx_rb.data(), h->hx_ro.data(), h->stream));
    return MGN_OK;
}

static int exchange_finish(mgn_handle* h) {
    COMMCHK(h, h->comm->a2a_finish(h->stream));
    if (!h->hx_direct) return mgn_halo_unpack(h, h->halo_recv.p);
    return MGN_OK;
}

// The schedule of one pass (the tested Python twin is engine.run_processor_staged): owned nodes are numbered boundary-first,
// so the boundary tiles are projected first, their P rows leave, and the interior projection plus the interior edge tiles of
// the next step run while the rows are on the wire; only the few boundary edge tiles wait for them.
static int project_and_start(mgn_handle* h, int k) {      // k = -1: projection for step 0
    if (int rc = mgn_proc_node_phase(h, k, 1)) return rc;
    if (int rc = exchange_start(h)) return rc;
    return mgn_proc_node_phase(h, k, 2);
}

static int processor_pass_staged(mgn_handle* h, int32_t nsteps, bool begin) {
    if (begin) {
        if (int rc = project_and_start(h, -1)) return rc;
    } else if (int rc = exchange_start(h)) return rc;      // P, Q of step 0 came from the encoder
    for (int k = 0; k < nsteps; ++k) {
        if (int rc = proc_edge_range(h, k, 1)) return rc;
        if (int rc = exchange_finish(h)) return rc;
        if (int rc = proc_edge_range(h, k, 2)) return rc;
        if (k + 1 < nsteps) {
            if (int rc = project_and_start(h, k)) return rc;
        } else if (int rc = mgn_proc_node(h, k, 0)) return rc;
    }
    return MGN_OK;
}

// every rank's (n_own, own_gid...) once per graph: output rows are gathered over the communicator
static int gather_plan(mgn_handle* h) {
    if (!h->all_gid.empty()) return MGN_OK;
    const LocalGraph& g = h->g;
    const int P = h->cfg.nranks;
    double mx = (double)g.n_own;
    COMMCHK(h, h->comm->allreduce_f64(&mx, 1, 1, h->stream));
    h->max_own = (int32_t)mx;
    const size_t per = (size_t)1 + h->max_own;
    std::vector<int32_t> mine(per, 0);
    mine[0] = g.n_own;
    memcpy(mine.data() + 1, g.own_gid.data(), (size_t)g.n_own * 4);
    HIPCHK(h, h->gath_s.ensure(per * 4));
    HIPCHK(h, h->gath_r.ensure(per * 4 * P));
    HIPCHK(h, hipMemcpyAsync(h->gath_s.p, mine.data(), per * 4, hipMemcpyHostToDevice, h->stream));
    COMMCHK(h, h->comm->allgather(h->gath_s.p, per * 4, h->gath_r.p, h->stream));
    std::vector<int32_t> all(per * P);
    HIPCHK(h, hipMemcpyAsync(all.data(), h->gath_r.p, per * 4 * P, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->all_gid.swap(all);
    return MGN_OK;
}

// local_dev [n_own][W] of every rank -> out [N][W] on the host, complete on every rank
static int gather_rows_global(mgn_handle* h, const float* local_dev, int W, float* out) {
    if (int rc = gather_plan(h)) return rc;
    const LocalGraph& g = h->g;
    const int P = h->cfg.nranks;
    const size_t per = (size_t)h->max_own * W * 4;
    HIPCHK(h, h->gath_s.ensure(per ? per : 16));
    HIPCHK(h, h->gath_r.ensure(per ? per * P : 16));
    HIPCHK(h, hipMemcpyAsync(h->gath_s.p, local_dev, (size_t)g.n_own * W * 4, hipMemcpyDeviceToDevice, h->stream));
    COMMCHK(h, h->comm->allgather(h->gath_s.p, per, h->gath_r.p, h->stream));
    std::vector<float> all((size_t)h->max_own * W * P);
    HIPCHK(h, hipMemcpyAsync(all.data(), h->gath_r.p, per * P, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t gper = (size_t)1 + h->max_own;
    for (int q = 0; q < P; ++q) {
        const int32_t* gq = h->all_gid.data() + (size_t)q * gper;
        const float* rows = all.data() + (size_t)q * h->max_own * W;
        for (int32_t i = 0; i < gq[0]; ++i) memcpy(out + (size_t)gq[1 + i] * W, rows + (size_t)i * W, (size_t)W * 4);
    }
    return MGN_OK;
}

// mgn.model(graph, ps, st) on a partitioned mesh: every rank encodes what it owns, the processor runs the staged schedule,
// the decoded rows are gathered so that every rank returns the complete output
static int forward_partitioned(mgn_handle* h, const float* nf, const float* ef, float* out) {
    if (int rc = need_comm(h, "mgn_forward")) return rc;
    if (int rc = need_set_features(h)) return rc;
    if (int rc = mgn_fwd_upload(h, nf, ef)) return rc;
    if (int rc = encode_impl(h, false)) return rc;
    if (int rc = processor_pass_staged(h, h->cfg.mps, false)) return rc;
    if (int rc = decode_impl(h, false)) return rc;
    return gather_rows_global(h, h->d_out.as<float>(), h->cfg.O, out);
}

static int processor_pass(mgn_handle* h, int32_t nsteps) {
    if (int rc = mgn_proc_begin(h)) return rc;
    return run_processor(h, nsteps);
}

// mgn_processor_steps_dev under ln_dims = MGN_LN_ALL (round 6): the resident latents leave their tiles as caller-order rows, the unfused
// whole-array driver (mgn_train.cpp: lnall_processor_steps; it takes device pointers) advances them, they go back.  Two row <-> tile passes
// per call on top of what mgn_processor_steps costs in this mode -- and no host copy.
static int lnall_processor_steps_dev(mgn_handle* h, int32_t nsteps) {
    if (int rc = need(h, true, true, true, true)) return rc;
    if (nsteps < 0 || nsteps > h->cfg.mps) return fail(h, MGN_E_ARG, "nsteps must be in [0, mps]");
    if (nsteps == 0) return MGN_OK;
    const LocalGraph& g = h->g;
    const EdgeTopo& t = g.set[0];
    auto& es = h->es[0];
    const int L = h->cfg.L;
    HIPCHK(h, h->lnall_v.ensure((size_t)(g.N > 0 ? g.N : 1) * L * 4));
    HIPCHK(h, h->lnall_e.ensure((size_t)(t.E > 0 ? t.E : 1) * L * 4));
    HIPCHK(h, launch_tiles_to_rows(h->V.as<float>(), nullptr, h->d_own_gid.as<int32_t>(), h->lnall_v.as<float>(), g.n_own, L, h->stream));
    if (t.e_local > 0)
        HIPCHK(h, launch_tiles_to_rows(es.Elat.as<float>(), es.d_edge_gid.as<int64_t>(), nullptr, h->lnall_e.as<float>(), t.e_local, L, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (int rc = lnall_processor_steps(h, h->lnall_v.as<float>(), h->lnall_e.as<float>(), nsteps)) return rc;   // (synchronises its stream)
    HIPCHK(h, launch_rows_to_tiles(h->lnall_v.as<float>(), nullptr, h->d_own_gid.as<int32_t>(), h->V.as<float>(), g.n_own, L, h->stream));
    if (t.e_local > 0)
        HIPCHK(h, launch_rows_to_tiles(h->lnall_e.as<float>(), es.d_edge_gid.as<int64_t>(), nullptr, es.Elat.as<float>(), t.e_local, L, h->stream));
    return MGN_OK;
}

int mgn_processor_steps_dev(mgn_handle* h, int32_t nsteps) try {
    if (h && !h->host_only && h->cfg.ln_dims == MGN_LN_ALL) return lnall_processor_steps_dev(h, nsteps);
    if (int rc = need(h, true, true)) return rc;
    if (nsteps < 0 || nsteps > h->cfg.mps) return fail(h, MGN_E_ARG, "nsteps must be in [0, mps]");
    if (nsteps == 0) return MGN_OK;
    if (h->cfg.nranks != 1 || (h->comm && h->force_staged)) {
        if (int rc = need_comm(h, "mgn_processor_steps_dev")) return rc;
        return processor_pass_staged(h, nsteps, true);
    }
    // Graph replay only helps launch-bound (small) passes; it needs a capturable stream (not the null stream) and no
    // per-launch event records.
#ifndef MGN_GRAPH_MAX_TILES
#define MGN_GRAPH_MAX_TILES 16384
#endif
    const bool graphable = h->use_graph && !h->prof && h->stream != nullptr && h->es[0].ntiles_e <= MGN_GRAPH_MAX_TILES;
    if (!graphable) return processor_pass(h, nsteps);
    if (h->graph_exec && h->graph_nsteps == nsteps) {
        HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
        return MGN_OK;
    }
    if (h->graph_warm != nsteps) {   // eager first: sets the per-kernel LDS attributes outside of any capture
        h->graph_warm = nsteps;
        return processor_pass(h, nsteps);
    }
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
    h->graph_nsteps = -1;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        h->use_graph = 0;
        return processor_pass(h, nsteps);
    }
    const int rc = processor_pass(h, nsteps);
    const hipError_t ce = hipStreamEndCapture(h->stream, &graph);
    if (rc != MGN_OK || ce != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        h->use_graph = 0;                       // fall back to eager launches for good
        if (rc != MGN_OK) return rc;
        return processor_pass(h, nsteps);
    }
    const hipError_t ie = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) {
        h->graph_exec = nullptr;
        h->use_graph = 0;
        return processor_pass(h, nsteps);
    }
    h->graph_nsteps = nsteps;
    HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
    return MGN_OK;
} MGN_CATCH(h)

int mgn_processor_steps(mgn_handle* h, float* v, float* e, int32_t nsteps) try {
    if (h && !h->host_only && h->cfg.ln_dims == MGN_LN_ALL) return lnall_processor_steps(h, v, e, nsteps);
    if (int rc = need(h, true, true)) return rc;
    if (h->cfg.nranks != 1) return fail(h, MGN_E_STATE, "mgn_processor_steps drives one partition");
    if (int rc = mgn_latents_import(h, v, e)) return rc;
    if (int rc = mgn_processor_steps_dev(h, nsteps)) return rc;
    return mgn_latents_export(h, v, e);
} MGN_CATCH(h)

// ---- halo ------------------------------------------------------------------------------------------
// one halo row carries the P row of every edge set: [set 0: L][set 1: L]
int mgn_halo_bytes_per_row(const mgn_handle* h) { return h ? h->nsets * h->cfg.L * (h->cfg.dtype == MGN_BF16 ? 2 : 4) : MGN_E_ARG; }

int mgn_halo_pack(mgn_handle* h, void* send_dev) try {
    if (int rc = need(h, false, true)) return rc;
    const int64_t rows = (int64_t)h->g.send_idx.size();
    if (rows == 0) return MGN_OK;
    if (!send_dev) return fail(h, MGN_E_ARG, "mgn_halo_pack: null buffer");
    ProfScope ps(h, F_HALO);
    const int L = h->cfg.L, stride = h->nsets * L;
    for (int q = 0; q < h->nsets; ++q) {
        if (is_bf16(h))
            HIPCHK(h, launch_gather_rows16(h->es[q].bP.as<uint16_t>(), h->d_send_idx.as<int32_t>(),
                                           reinterpret_cast<uint16_t*>(send_dev) + (size_t)q * L, rows, stride, h->stream));
        else
            HIPCHK(h, launch_gather_rows(h->es[q].P.as<float>(), h->d_send_idx.as<int32_t>(), reinterpret_cast<float*>(send_dev) + (size_t)q * L,
                                         rows, L, stride, h->stream));
    }
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_halo_unpack(mgn_handle* h, const void* recv_dev) try {
    if (int rc = need(h, false, true)) return rc;
    const LocalGraph& g = h->g;
    if (g.n_halo == 0) return MGN_OK;
    if (!recv_dev) return fail(h, MGN_E_ARG, "mgn_halo_unpack: null buffer");
    ProfScope ps(h, F_HALO);
    const size_t b = is_bf16(h) ? 2 : 4, rowb = (size_t)h->cfg.L * b;
    for (int q = 0; q < h->nsets; ++q) {
        if (is_bf16(h) && prows_blocked()) {
            HIPCHK(h, launch_scatter_prows16(reinterpret_cast<const uint16_t*>(recv_dev) + (size_t)q * h->cfg.L, h->nsets * h->cfg.L,
                                             h->es[q].bP.as<uint16_t>(), g.n_own, g.n_halo, h->stream));
            continue;
        }
        if (!is_bf16(h) && prows_blocked()) {
            HIPCHK(h, launch_scatter_prows(reinterpret_cast<const float*>(recv_dev) + (size_t)q * h->cfg.L, h->nsets * h->cfg.L, h->es[q].P.as<float>(),
                                           g.n_own, g.n_halo, h->cfg.L, h->stream));
            continue;
        }
        char* dst = reinterpret_cast<char*>(is_bf16(h) ? h->es[q].bP.p : h->es[q].P.p) + (size_t)g.n_own * rowb;
        HIPCHK(h, hipMemcpy2DAsync(dst, rowb, reinterpret_cast<const char*>(recv_dev) + (size_t)q * rowb, (size_t)h->nsets * rowb, rowb,
                                   (size_t)g.n_halo, hipMemcpyDeviceToDevice, h->stream));
    }
    return MGN_OK;
} MGN_CATCH(h)



// ---- device-side graph prologue (reference create_base_graph, src/graph.jl:25-55) ----------------------------
int mgn_triangles_to_edges_dev(mgn_handle* h, const int32_t* cells, int64_t n_cells, int32_t* senders, int32_t* receivers, int64_t capacity,
                               int64_t* n_directed) try {
    if (int rc = need(h, false, false)) return rc;
    if (!cells || n_cells < 0 || !n_directed || capacity < 0 || (capacity > 0 && (!senders || !receivers)))
        return fail(h, MGN_E_ARG, "mgn_triangles_to_edges_dev: bad argument");
    *n_directed = 0;
    if (n_cells == 0) return MGN_OK;
    const size_t cb = (size_t)n_cells * 3 * 4;
    HIPCHK(h, h->gpos.ensure(cb));
    HIPCHK(h, hipMemcpyAsync(h->gpos.p, cells, cb, hipMemcpyDefault, h->stream));      // host or device source
    HIPCHK(h, h->gout.ensure((size_t)(capacity > 0 ? capacity : 1) * 8));
    int32_t* ds = h->gout.as<int32_t>();
    int32_t* dr = ds + (capacity > 0 ? capacity : 1);
    int64_t m = 0;
    const hipError_t e = dev_triangles_to_edges(h->gpos.as<int32_t>(), n_cells, h->gwork, ds, dr, capacity, &m, h->stream);
    *n_directed = 2 * m;
    if (e == hipErrorInvalidValue && 2 * m > capacity)
        return fail(h, MGN_E_ARG, "mgn_triangles_to_edges_dev: %lld directed edges do not fit the buffers (capacity %lld; 6 x n_cells always does)",
                    (long long)(2 * m), (long long)capacity);
    HIPCHK(h, e);
    HIPCHK(h, hipMemcpyAsync(senders, ds, (size_t)(2 * m) * 4, hipMemcpyDefault, h->stream));
    HIPCHK(h, hipMemcpyAsync(receivers, dr, (size_t)(2 * m) * 4, hipMemcpyDefault, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MGN_OK;
} MGN_CATCH(h)

int mgn_set_static_mesh(mgn_handle* h, const int32_t* node_type, int32_t type_min, int32_t type_max, const float* mesh_pos, int32_t pos_dim,
                        const float* val_mask) try {
    if (int rc = need(h, true, true)) return rc;
    const mgn_config& c = h->cfg;
    if (c.nranks != 1) return fail(h, MGN_E_STATE, "mgn_set_static_mesh drives one partition");
    if (h->nsets != 1) return fail(h, MGN_E_STATE, "mgn_set_static_mesh mirrors the reference's single-edge-set create_base_graph (src/graph.jl:25-55)");
    if (!node_type || !mesh_pos || type_max < type_min) return fail(h, MGN_E_ARG, "mgn_set_static_mesh: bad argument");
    const int depth = type_max - type_min + 1;
    if (c.Fn - c.O != depth) return fail(h, MGN_E_ARG, "mgn_set_static_mesh: Fn - O = %d but the one-hot depth is %d", c.Fn - c.O, depth);
    if (c.Fe != pos_dim + 1) return fail(h, MGN_E_ARG, "mgn_set_static_mesh: Fe = %d but mesh_pos has %d columns (Fe = dims + 1, src/graph.jl:49-52)", c.Fe, pos_dim);
    const LocalGraph& g = h->g;
    invalidate_static(h);
    h->in_wa = c.O;
    h->in_wb = depth;
    h->in_local = true;          // the static inputs are produced in ENGINE order (owned nodes, receiver-sorted local edges)
    HIPCHK(h, h->d_nfA.ensure((size_t)g.N * c.O * 4));
    HIPCHK(h, h->d_nfB.ensure((size_t)g.n_own * depth * 4));
    HIPCHK(h, h->es[0].d_ef.ensure((size_t)(g.set[0].e_local > 0 ? g.set[0].e_local : 1) * c.Fe * 4));
    HIPCHK(h, h->gtype.ensure((size_t)g.N * 4));
    HIPCHK(h, h->gpos.ensure((size_t)g.N * pos_dim * 4));
    HIPCHK(h, hipMemcpyAsync(h->gtype.p, node_type, (size_t)g.N * 4, hipMemcpyDefault, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->gpos.p, mesh_pos, (size_t)g.N * pos_dim * 4, hipMemcpyDefault, h->stream));
    // one partition: no boundary block; local ids are global ids unless mgn_set_graph renumbered the mesh (then through own_gid)
    const int32_t* l2g = g.renumbered ? h->d_own_gid.as<int32_t>() : nullptr;
    HIPCHK(h, launch_one_hot(h->gtype.as<int32_t>(), l2g, g.n_own, type_min, depth, h->d_nfB.as<float>(), h->stream));
    HIPCHK(h, launch_edge_features_local(h->gpos.as<float>(), pos_dim, h->es[0].d_snd.as<int32_t>(), h->es[0].d_rcv.as<int32_t>(), l2g,
                                         g.set[0].e_local, h->es[0].d_ef.as<float>(), h->stream));
    h->have_mask = val_mask != nullptr;
    if (val_mask) {
        HIPCHK(h, h->d_mask.ensure((size_t)g.N * 4));
        HIPCHK(h, hipMemcpyAsync(h->d_mask.p, val_mask, (size_t)g.N * 4, hipMemcpyDefault, h->stream));
    }
    if (int rc = encode_impl(h, true, false, true)) return rc;       // edge encoder: once per trajectory
    const bool bf = is_bf16(h);
    const size_t eb = tile_floats(h->es[0].ntiles_e, c.L) * (bf ? 2 : 4);
    HIPCHK(h, h->es[0].elat0.ensure(eb));
    HIPCHK(h, hipMemcpyAsync(h->es[0].elat0.p, bf ? h->es[0].bElat.p : h->es[0].Elat.p, eb, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_static = true;
    return MGN_OK;
} MGN_CATCH(h)

int mgn_world_edges_dev(mgn_handle* h, int32_t set, const float* world_pos, int32_t dim, float radius, int64_t* n_edges) try {
    if (int rc = need(h, false, true)) return rc;
    if (h->cfg.nranks != 1) return fail(h, MGN_E_STATE, "mgn_world_edges_dev drives one partition (with nranks > 1 search on the host: mgn_world_edges + mgn_set_edge_set)");
    if (set < 1 || set >= h->nsets) return fail(h, MGN_E_ARG, "mgn_world_edges_dev: set %d out of range (handle has %d edge sets)", set, h->nsets);
    if (!world_pos || dim < 1 || dim > 3 || !(radius > 0.f)) return fail(h, MGN_E_ARG, "mgn_world_edges_dev: bad argument");
    LocalGraph& g = h->g;
    auto& es = h->es[set];
    h->hx_ready = false;
    invalidate_static(h);
    train_invalidate(h, 2);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_graph(h);
    HIPCHK(h, h->gpos.ensure((size_t)g.N * dim * 4));
    if (g.renumbered) {   // the search runs in the engine's node numbering (set 0's adjacency is held in it): positions gathered through own_gid
        HIPCHK(h, h->stage.ensure((size_t)g.N * dim * 4));
        HIPCHK(h, hipMemcpyAsync(h->stage.p, world_pos, (size_t)g.N * dim * 4, hipMemcpyDefault, h->stream));
        HIPCHK(h, launch_permute_rows(h->gpos.as<float>(), h->stage.as<float>(), h->d_own_gid.as<int32_t>(), g.N, dim, false, h->stream));
    } else {
        HIPCHK(h, hipMemcpyAsync(h->gpos.p, world_pos, (size_t)g.N * dim * 4, hipMemcpyDefault, h->stream));
    }
    int64_t E = 0;
    const hipError_t e = dev_world_edges(h->gpos.as<float>(), dim, g.N, radius, h->es[0].d_rowptr.as<int32_t>(), h->es[0].d_snd.as<int32_t>(), h->gwork,
                                         es.d_snd, es.d_rcv, es.d_rowptr, &E, h->stream);
    if (e == hipErrorInvalidValue) return fail(h, MGN_E_ARG, "mgn_world_edges_dev: non-finite position");
    HIPCHK(h, e);
    // the set lives on the device only (receiver-major == engine order, global edge id == position): no host lists are kept
    EdgeTopo& t = g.set[set];
    t.E = t.e_local = E;
    t.halo_span = 0;
    t.snd.clear(); t.rcv.clear(); t.rowptr.clear(); t.edge_gid.clear();
    es.gs.clear(); es.gr.clear();
    es.ntiles_e = (int32_t)((E + TILE - 1) / TILE);
    HIPCHK(h, es.d_edge_gid.ensure((size_t)(E > 0 ? E : 1) * 8));
    HIPCHK(h, launch_iota64(es.d_edge_gid.as<int64_t>(), E, h->stream));
    if (int rc = alloc_edge_set(h, set)) return rc;
    es.have_ef = false;
    if (es.Fe == dim + 1) {      // [rel world pos ; norm]: the world-edge features of the cloth models, straight into the encoder's input
        HIPCHK(h, es.d_ef.ensure((size_t)(E > 0 ? E : 1) * es.Fe * 4));
        HIPCHK(h, launch_edge_features_local(h->gpos.as<float>(), dim, es.d_snd.as<int32_t>(), es.d_rcv.as<int32_t>(), nullptr, E, es.d_ef.as<float>(), h->stream));
        es.have_ef = true;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_edges) *n_edges = E;
    return MGN_OK;
} MGN_CATCH(h)

/* the device-built set's topology, for hosts that want it (tests): senders / receivers [E] 0-based, receiver-major */
int mgn_edge_set_export(mgn_handle* h, int32_t set, int32_t* senders, int32_t* receivers) try {
    if (int rc = need(h, false, true)) return rc;
    if (set < 0 || set >= h->nsets) return fail(h, MGN_E_ARG, "mgn_edge_set_export: set %d out of range", set);
    if (h->cfg.nranks != 1) return fail(h, MGN_E_STATE, "mgn_edge_set_export drives one partition");
    const int64_t E = h->g.set[set].e_local;
    if (E > 0 && (!senders || !receivers)) return fail(h, MGN_E_ARG, "mgn_edge_set_export: null output");
    // every array crosses the boundary in the CALLER's node numbering: a renumbered handle (graph_host.h) holds engine-order ids and maps
    // them back on the host -- so it is decided BEFORE anything is written whether the outputs can be walked by the host (a device
    // output on a renumbered handle is refused with its buffers untouched; an output the runtime cannot classify counts as device memory
    // unless it is plain host memory the runtime has never seen)
    if (h->g.renumbered && E > 0) {
        for (const int32_t* p : {senders, receivers}) {
            hipPointerAttribute_t at{};
            const hipError_t e = hipPointerGetAttributes(&at, p);
            (void)hipGetLastError();
            const bool host_walkable = e == hipErrorInvalidValue /* unregistered host memory */ ||
                                       (e == hipSuccess && (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeUnregistered || at.type == hipMemoryTypeManaged));
            if (!host_walkable) return fail(h, MGN_E_UNSUPPORTED, "mgn_edge_set_export: device output on a renumbered handle (pass host arrays)");
        }
    }
    if (E > 0) {
        HIPCHK(h, hipMemcpyAsync(senders, h->es[set].d_snd.p, (size_t)E * 4, hipMemcpyDefault, h->stream));
        HIPCHK(h, hipMemcpyAsync(receivers, h->es[set].d_rcv.p, (size_t)E * 4, hipMemcpyDefault, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->g.renumbered && E > 0) {
        for (int64_t j = 0; j < E; ++j) {
            senders[j] = h->g.own_gid[(size_t)senders[j]];
            receivers[j] = h->g.own_gid[(size_t)receivers[j]];
        }
    }
    return MGN_OK;
} MGN_CATCH(h)

// ---- communicator entry points ---------------------------------------------------------------------------------------
int mgn_comm_unique_id(void* id, int32_t transport) try {
    if (!id) return fail(nullptr, MGN_E_ARG, "mgn_comm_unique_id: null id");
    std::string why;
    if (comm_unique_id(id, transport, why) != 0) return fail(nullptr, MGN_E_RCCL, "mgn_comm_unique_id: %s", why.c_str());
    return MGN_OK;
} MGN_CATCH(nullptr)

int mgn_comm_init(mgn_handle* h, const void* id, size_t id_bytes, int32_t transport) try {
    if (!h) return MGN_E_ARG;
    if (!id || id_bytes != MGN_COMM_ID_BYTES) return fail(h, MGN_E_ARG, "mgn_comm_init: id must be MGN_COMM_ID_BYTES (%d) bytes", MGN_COMM_ID_BYTES);
    if (transport != MGN_COMM_RCCL && transport != MGN_COMM_HOST) return fail(h, MGN_E_ARG, "mgn_comm_init: unknown transport %d", transport);
    if (h->comm) return fail(h, MGN_E_STATE, "mgn_comm_init: the handle already has a communicator (mgn_comm_destroy first)");
    if (!h->host_only) HIPCHK(h, hipStreamSynchronize(h->stream));
    std::string why;
    h->comm = comm_create(id, transport, h->cfg.rank, h->cfg.nranks, !h->host_only, why);
    if (!h->comm) return fail(h, MGN_E_RCCL, "mgn_comm_init: %s", why.c_str());
    if (const char* e = getenv("MGN_FORCE_STAGED")) h->force_staged = atoi(e);
    h->hx_ready = false;
    h->all_gid.clear();
    if (!h->host_only) drop_graph(h);
    return MGN_OK;
} MGN_CATCH(h)

// File bootstrap of the communicator id (single node, no launcher store).  A file left behind by an earlier run (or a crash) must
// never be taken for this run's id, so the id travels with a nonce and is confirmed both ways:
//   rank 0: removes path, path.go and stale acks; writes path = {id, nonce}; waits until every other rank has written
//           path.ack.<rank> = nonce; writes path.go = nonce; after the collective mgn_comm_init has returned removes every file.
//   rank r: reads path, writes its ack with the nonce it read, waits for a path.go that carries the same nonce AND is not older
//           than its own ack (a stale go predates it); on a different nonce it starts over with a fresh read of path.
namespace {
struct IdFile { unsigned char id[MGN_COMM_ID_BYTES]; uint64_t nonce; };
bool write_atomic(const std::string& path, const void* data, size_t n) {
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(data, 1, n, f) == n;
    fclose(f);
    return ok && rename(tmp.c_str(), path.c_str()) == 0;
}
bool read_exact(const std::string& path, void* data, size_t n) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t got = fread(data, 1, n, f);
    fclose(f);
    return got == n;
}
double mtime_of(const std::string& path) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return -1.0;
    return (double)st.st_mtim.tv_sec + 1e-9 * (double)st.st_mtim.tv_nsec;
}
void nap20ms() {
    struct timespec ts = {0, 20 * 1000 * 1000};
    nanosleep(&ts, nullptr);
}
}  // namespace

int mgn_comm_init_file(mgn_handle* h, const char* path, int32_t transport) try {
    if (!h || !path) return fail(h, MGN_E_ARG, "mgn_comm_init_file: null argument");
    const double limit = getenv("MGN_COMM_TIMEOUT_S") ? atof(getenv("MGN_COMM_TIMEOUT_S")) : 120.0;
    const std::string base(path), go = base + ".go";
    const int rank = h->cfg.rank, nranks = h->cfg.nranks;
    auto ack_of = [&](int r) { return base + ".ack." + std::to_string(r); };
    IdFile rec{};
    if (rank == 0) {
        unlink(path);
        unlink(go.c_str());
        for (int r = 1; r < nranks; ++r) unlink(ack_of(r).c_str());
        if (int rc = mgn_comm_unique_id(rec.id, transport)) return fail(h, rc, "%s", mgn_last_error(nullptr));
        struct timespec now;
        clock_gettime(CLOCK_REALTIME, &now);
        rec.nonce = ((uint64_t)now.tv_sec << 32) ^ ((uint64_t)now.tv_nsec << 8) ^ (uint64_t)getpid();
        FILE* ur = fopen("/dev/urandom", "rb");
        if (ur) { uint64_t x = 0; if (fread(&x, 1, 8, ur) == 8) rec.nonce ^= x; fclose(ur); }
        if (!write_atomic(base, &rec, sizeof rec)) return fail(h, MGN_E_RCCL, "mgn_comm_init_file: cannot write %s", path);
        double waited = 0;
        for (int r = 1; r < nranks; ++r)
            for (;;) {
                uint64_t n = 0;
                if (read_exact(ack_of(r), &n, 8) && n == rec.nonce) break;
                if (waited > limit) return fail(h, MGN_E_RCCL, "mgn_comm_init_file: timed out waiting for rank %d to acknowledge %s", r, path);
                nap20ms();
                waited += 0.02;
            }
        if (!write_atomic(go, &rec.nonce, 8)) return fail(h, MGN_E_RCCL, "mgn_comm_init_file: cannot write %s", go.c_str());
    } else {
        double waited = 0;
        for (bool confirmed = false; !confirmed;) {
            while (!read_exact(base, &rec, sizeof rec)) {
                if (waited > limit) return fail(h, MGN_E_RCCL, "mgn_comm_init_file: timed out waiting for rank 0 to write %s", path);
                nap20ms();
                waited += 0.02;
            }
            if (!write_atomic(ack_of(rank), &rec.nonce, 8)) return fail(h, MGN_E_RCCL, "mgn_comm_init_file: cannot write %s", ack_of(rank).c_str());
            const double t_ack = mtime_of(ack_of(rank));
            for (;;) {
                uint64_t n = 0;
                if (read_exact(go, &n, 8) && mtime_of(go) >= t_ack) {
                    confirmed = n == rec.nonce;
                    break;                                   // a go for another nonce: what this rank read was stale -> read again
                }
                IdFile cur{};
                if (read_exact(base, &cur, sizeof cur) && cur.nonce != rec.nonce) break;   // rank 0 has published a new id meanwhile
                if (waited > limit) return fail(h, MGN_E_RCCL, "mgn_comm_init_file: timed out waiting for rank 0 to confirm %s", path);
                nap20ms();
                waited += 0.02;
            }
            if (!confirmed) { nap20ms(); waited += 0.02; }
        }
    }
    const int rc = mgn_comm_init(h, rec.id, MGN_COMM_ID_BYTES, transport);
    if (rank == 0) {                                         // the collective init is over on every rank that could reach it
        unlink(path);
        unlink(go.c_str());
        for (int r = 1; r < nranks; ++r) unlink(ack_of(r).c_str());
    }
    return rc;
} MGN_CATCH(h)

int mgn_comm_destroy(mgn_handle* h) try {
    if (!h) return MGN_E_ARG;
    if (!h->host_only) (void)hipStreamSynchronize(h->stream);
    delete h->comm;
    h->comm = nullptr;
    h->hx_ready = false;
    h->all_gid.clear();
    return MGN_OK;
} MGN_CATCH(h)

int mgn_comm_barrier(mgn_handle* h) try {
    if (!h) return MGN_E_ARG;
    if (int rc = need_comm(h, "mgn_comm_barrier")) return rc;
    COMMCHK(h, h->comm->barrier(h->stream));
    return MGN_OK;
} MGN_CATCH(h)

int mgn_comm_allreduce(mgn_handle* h, double* x, int32_t n, int32_t op) try {
    if (!h) return MGN_E_ARG;
    if (int rc = need_comm(h, "mgn_comm_allreduce")) return rc;
    if (!x || n < 0 || (op != 0 && op != 1)) return fail(h, MGN_E_ARG, "mgn_comm_allreduce: bad argument");
    COMMCHK(h, h->comm->allreduce_f64(x, n, op, h->stream));
    return MGN_OK;
} MGN_CATCH(h)

int mgn_halo_exchange(mgn_handle* h) try {
    if (int rc = need(h, false, true)) return rc;
    if (int rc = need_comm(h, "mgn_halo_exchange")) return rc;
    if (int rc = exchange_start(h)) return rc;
    return exchange_finish(h);
} MGN_CATCH(h)

int mgn_halo_exchange_host(mgn_handle* h, const float* own_rows, float* halo_rows, int32_t width) try {
    if (!h || !h->have_graph) return fail(h, MGN_E_STATE, "mgn_halo_exchange_host before mgn_set_graph");
    if (int rc = need_comm(h, "mgn_halo_exchange_host")) return rc;
    if (width < 1 || (!own_rows && h->g.n_own) || (!halo_rows && h->g.n_halo)) return fail(h, MGN_E_ARG, "mgn_halo_exchange_host: bad argument");
    const LocalGraph& g = h->g;
    const int P = h->cfg.nranks;
    const size_t rowb = (size_t)width * 4;
    std::vector<float> send(g.send_idx.size() * (size_t)width);
    for (size_t i = 0; i < g.send_idx.size(); ++i) memcpy(send.data() + i * width, own_rows + (size_t)g.send_idx[i] * width, rowb);
    std::vector<size_t> sb(P), so(P), rb(P), ro(P);
    size_t s0 = 0, r0 = 0;
    for (int q = 0; q < P; ++q) {
        sb[q] = (size_t)g.send_rows[q] * rowb; so[q] = s0; s0 += sb[q];
        rb[q] = (size_t)g.recv_rows[q] * rowb; ro[q] = r0; r0 += rb[q];
    }
    COMMCHK(h, h->comm->a2a_host(send.data(), sb.data(), so.data(), halo_rows, rb.data(), ro.data()));
    return MGN_OK;
} MGN_CATCH(h)

// Process-wide kernel-path override for tests (not part of the public header): 0 auto, 1 LDS-resident persistent
// kernels, 2 all-streaming, 3 cooperative 4-wave tiles.  Returns the previous value.
int mgn_debug_kernel_path(int path) { return set_kernel_path(path); }
int mgn_debug_c16_row_tiles(int rt) { return set_c16_row_tiles(rt); }
int mgn_debug_c16_edge_tiles(int t) { return set_c16_edge_tiles(t); }   // size limit of the 16-row kernels in edge tiles per CU (0: by the handle)
// large fp32 launches: 1 split path (k_edge_ring + k_node_split + k_project_split: bf16 matrix cores at fp32 accuracy; the default),
// 0 fp32-MFMA kernels; returns the old value
int mgn_debug_fp32_split(int on) { return set_fp32_split(on); }
// 1 (default): the split path computes on two fp16 pieces per operand and three piece products (k_edge_ring_h), 0: on three bf16 pieces
// and six products (k_edge_ring); returns the old value
int mgn_debug_split_f16(int on) { return set_split_f16(on); }
// the same switch for the streaming kernels of the training step (train.hip: train_chunk); returns the old value
int mgn_debug_train_f16(int on) { return set_train_f16(on); }
// tests: every chunk the device packed (k_pack_weights) against the host functions that specify the layouts; returns the number of
// elements that differ (0 = bitwise equal), < 0 on error
long long mgn_debug_pack_check(mgn_handle* h) try {
    if (int rc = need(h, true, false)) return -rc;
    const int L = h->cfg.L;
    const size_t CH = (size_t)L * L;
    const float* p = h->params.data();
    long long bad = 0;
    std::vector<float> hf(3 * CH), df(3 * CH);
    std::vector<uint16_t> hb(3 * 16384), db(3 * 16384);
    for (const WPackJob& jb : h->wjobs) {
        const float* W = p + jb.src;
        if (jb.kind == 0) {
            const size_t n = (L == 128 ? 3 : 2) * CH;
            pack_chunk(hf.data(), W, jb.ldw, jb.kbase, L);
            pack_chunk_tmajor(hf.data() + CH, hf.data(), L);
            if (L == 128) pack_chunk16(hf.data() + 2 * CH, W, jb.ldw, jb.kbase);
            if (hipMemcpy(df.data(), h->wfrag.as<float>() + jb.off, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -MGN_E_HIP;
            for (size_t i = 0; i < n; ++i) bad += memcmp(&hf[i], &df[i], 4) != 0;
        } else if (jb.kind == 3) {
            pack_chunk_bf16(hb.data(), W, jb.ldw, jb.kbase);
            if (hipMemcpy(db.data(), h->wbf.as<uint16_t>() + jb.off, 16384 * 2, hipMemcpyDeviceToHost) != hipSuccess) return -MGN_E_HIP;
            for (size_t i = 0; i < 16384; ++i) bad += hb[i] != db[i];
        } else if (jb.kind == 4 || jb.kind == 5) {
            pack_chunk_split_h(hb.data(), W, jb.ldw, jb.kbase, jb.scale, jb.kind == 5);
            if (hipMemcpy(db.data(), h->wsp.as<uint16_t>() + jb.off, 2 * 16384 * 2, hipMemcpyDeviceToHost) != hipSuccess) return -MGN_E_HIP;
            for (size_t i = 0; i < 2 * 16384; ++i) bad += hb[i] != db[i];
        } else {
            pack_chunk_split(hb.data(), W, jb.ldw, jb.kbase, jb.kind == 2);
            if (hipMemcpy(db.data(), h->wsp.as<uint16_t>() + jb.off, 3 * 16384 * 2, hipMemcpyDeviceToHost) != hipSuccess) return -MGN_E_HIP;
            for (size_t i = 0; i < 3 * 16384; ++i) bad += hb[i] != db[i];
        }
    }
    return bad;
} catch (...) { return -MGN_E_OOM; }
int mgn_debug_c16_split(int on) { return set_c16_split(on); }   // bits: 1 edge kernel (default), 2 node kernel, 4 edge kernel at one row tile per block; 0: fp32 MFMA pipe (kernels.hip)
int mgn_debug_last_node_kernel(void) { return last_node_kernel(); }   // the same for the node MLP (codes: kernels.hip, launch_node_step)
int mgn_debug_last_edge_kernel(void) { return last_edge_kernel(); }   // kernels.hip: which family the last fp32 edge launch ran on
// node numbering policy of the NEXT mgn_set_graph calls (0 never, 1 auto, 2 always breadth-first); returns the old value
int mgn_debug_renumber(int mode) { const int old = g_renumber; g_renumber = mode; return old; }
int mgn_debug_renumbered(const mgn_handle* h) { return h && h->have_graph && h->g.renumbered ? 1 : 0; }

int mgn_debug_edge_stamps(mgn_handle* h, int32_t k, unsigned long long* out /* [32768] */) try {
    if (int rc = need(h, true, true)) return rc;
    const size_t n = 32768;
    HIPCHK(h, h->d_stamps.ensure(n * 8));
    HIPCHK(h, hipMemsetAsync(h->d_stamps.p, 0, n * 8, h->stream));
    if (int rc = mgn_proc_edge(h, k)) return rc;
    HIPCHK(h, hipMemcpyAsync(out, h->d_stamps.p, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->d_stamps.release();
    return MGN_OK;
} MGN_CATCH(h)

int mgn_debug_node_stamps(mgn_handle* h, int32_t k, unsigned long long* out /* [32768] */) try {
    if (int rc = need(h, true, true)) return rc;
    const size_t n = 32768;
    HIPCHK(h, h->d_stamps.ensure(n * 8));
    HIPCHK(h, hipMemsetAsync(h->d_stamps.p, 0, n * 8, h->stream));
    if (int rc = mgn_proc_node(h, k, 1)) return rc;
    HIPCHK(h, hipMemcpyAsync(out, h->d_stamps.p, n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->d_stamps.release();
    return MGN_OK;
} MGN_CATCH(h)

// ---- profiling ---------------------------------------------------------------------------------------
int mgn_profile_enable(mgn_handle* h, int32_t on) try {
    if (int rc = need(h, false, false)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (auto& r : h->recs) {
        h->event_pool.push_back(r.a);
        h->event_pool.push_back(r.b);
    }
    h->recs.clear();
    h->prof = on != 0;
    // events are created here, outside of whatever the caller is about to time (one per scope boundary; the pool grows on
    // demand beyond this)
    while (h->prof && h->event_pool.size() < 4096) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) break;
        h->event_pool.push_back(e);
    }
    return MGN_OK;
} MGN_CATCH(h)

int mgn_profile_read(mgn_handle* h, double ms_avg[8], int64_t counts[8]) try {
    if (!h || !ms_avg || !counts) return MGN_E_ARG;
    if (int rc = need(h, false, false)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double tot[8] = {0};
    int64_t cnt[8] = {0};
    for (auto& r : h->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            tot[r.fam] += ms;
            cnt[r.fam] += 1;
        }
        h->event_pool.push_back(r.a);
        h->event_pool.push_back(r.b);
    }
    h->recs.clear();
    for (int i = 0; i < 8; ++i) {
        ms_avg[i] = cnt[i] ? tot[i] / (double)cnt[i] : 0.0;
        counts[i] = cnt[i];
    }
    return MGN_OK;
} MGN_CATCH(h)

}  // extern "C"
