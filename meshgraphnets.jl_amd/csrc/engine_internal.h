// Engine state shared by the C-ABI translation units (mgn_api.cpp: inference path, mgn_train.cpp: step!).
// Internal; the public boundary is include/mgn_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/mgn_hip.h"
#include "comm.h"
#include "graph_host.h"
#include "kernels.h"
#include "train.h"

namespace mgn {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// offsets (in floats) of one MLP inside the packed parameter vector: nl = hidden_layers + 1 Dense layers
constexpr int MAX_DENSE = 5;        // hidden_layers <= 4
struct MlpOff {
    size_t W[MAX_DENSE], b[MAX_DENSE], gamma = 0, beta = 0;
    int in = 0, out = 0, nl = 3;
    bool ln = false;
};
// offsets (floats, into wfrag) of the Dense layers after the first one of an MLP, for the GEN kernels (kernels.h GenMlp)
struct GenOff {
    size_t ch[4] = {0, 0, 0, 0}, tabs = 0;
};

enum Family { F_EDGE = 0, F_NODE, F_ENC, F_DEC, F_HALO, F_EDGE_BND, F_NFAM };   // F_EDGE_BND: boundary tiles of a split edge step

struct ProfRec {
    int fam;
    hipEvent_t a, b;
};

}  // namespace mgn

namespace mgn { struct TrainState; }
using mgn::DevBuf;
using mgn::MlpOff;
using mgn::GenOff;
using mgn::ProfRec;
using mgn::LocalGraph;
using mgn::MAX_EDGE_SETS;

struct mgn_engine {
    // (fields use mgn:: types)

    mgn_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    bool host_only = false;
    int32_t node_split = 1;   // projection as its own launch (both chunks LDS-resident); MGN_NODE_SPLIT=0 fuses it
    int32_t stagger_edge = 0, stagger_node = 8;  // tunables (MGN_STAGGER_EDGE / MGN_STAGGER_NODE); edge: 0 since the padded MFMAs (sweep 0..64: 4.558 .. 4.621 ms)

    // parameters
    bool have_params = false;
    DevBuf d_params, d_wjobs;   // the parameter vector and the chunk list on the device: what k_pack_weights builds the layouts from
    std::vector<mgn::WPackJob> wjobs;   // (the list, kept for mgn_debug_pack_check)
    bool packed_ok = false;     // the kernels' weight layouts (wfrag, wsp, wbf) are those of `params`: mgn_set_params only stores the vector,
                                // the first compute call that reads them packs (a training loop -- set_params, step!, ... -- never does)
    std::vector<float> params;  // packed, host
    MlpOff enc_node, dec;
    std::vector<MlpOff> pn;
    DevBuf wfrag;     // all chunks + tables + small tensors, fragment order
    // offsets into wfrag (floats).  e_ch / e_tabs: edge MLP of each set; n_ch: node MLP (0:W2 1:W3 2:W1v 3:W1a 6:W1a of
    // set 1) and the projection onto the NEXT step's set-0 edge MLP (4:WP 5:WQ, bias in n_tabs[T_BQ]); p1_ch / p1_tabs:
    // the same projection for set 1
    struct StepOff { size_t e_ch[MAX_EDGE_SETS][3], e_tabs[MAX_EDGE_SETS], n_ch[7], n_tabs, p1_ch[2], p1_tabs; GenOff e_gen[MAX_EDGE_SETS], n_gen; };
    std::vector<StepOff> soff;
    size_t en_ch[4] = {0, 0, 0, 0}, en_tabs = 0, en_w1f = 0;
    size_t de_ch[2] = {0, 0}, de_tabs = 0, de_w3f = 0, de_b3 = 0;
    GenOff en_gen, de_gen;

    // norms (device): node scale/shift [Fn], edge [Fe], out [O]; null = identity
    DevBuf norms;
    std::vector<float> norms_host;   // the same affine maps on the host (the whole-array LayerNorm mode's mgn_ode_step builds its inputs there)
    bool have_nnorm = false, have_enorm = false, have_onorm = false;

    // graph
    bool have_graph = false;
    LocalGraph g;
    int32_t nsets = 1;
    int32_t ntiles_n = 0;
    DevBuf d_own_gid, d_send_idx;
    // per edge set: parameters, topology, latents (set 0 = the reference's mesh edges; set 1 = world edges)
    struct EdgeSetState {
        int32_t Fe = 0;
        MlpOff enc;
        std::vector<MlpOff> pe;
        size_t ee_ch[2] = {0, 0}, ee_tabs = 0, ee_w1f = 0;
        GenOff ee_gen;
        int32_t ntiles_e = 0;
        bool have_ef = false;
        DevBuf d_snd, d_rcv, d_rowptr, d_edge_gid, d_ef;
        DevBuf Elat, AGG, CARRY, P, Q, elat0;
        DevBuf bP, bQ, bElat, bAGG, bCARRY;   // bf16 mode
        std::vector<int32_t> gs, gr;          // host copy of the global edge list (kept only with two sets: rebuilds)
        int32_t gbase = 0;
    } es[MAX_EDGE_SETS];

    // latents and I/O
    // bf16 mode (cfg.dtype == MGN_BF16): bf16 copies of the processor state and weights; the fp32 V / Elat buffers
    // then only carry encoder output / decoder input
    DevBuf wbf, bV;
    struct BfStepOff { size_t e_ch[MAX_EDGE_SETS][3], n_ch[7], p1_ch[2]; };
    std::vector<BfStepOff> bsoff;
    // split path (csrc/split.hip: fp32 storage, products on the bf16 matrix cores): per step and set the three edge chunks, per step the
    // node MLP's four and the projection's two, as 3 bf16 pieces each
    DevBuf wsp;
    struct SplitOff { size_t e_ch[MAX_EDGE_SETS][3], e16_ch[MAX_EDGE_SETS][3]; size_t n_ch[6], n2_ch[3], n16_ch[9]; bool have_n;
                      // two fp16 pieces (kind 4 of WPackJob): offsets, the power of two each chunk was multiplied by, max(0, max b2)
                      size_t eh_ch[MAX_EDGE_SETS][3], nh_ch[6], e16h_ch[MAX_EDGE_SETS][3], n16h_ch[9]; float eh_s[MAX_EDGE_SETS][3], nh_s[9], e_b2pos[MAX_EDGE_SETS], n_b2pos; bool have_h; };   // n2_ch (two edge sets): W1[2L:3L], WP / WQ of set 1
    std::vector<SplitOff> spoff;
    // encoders / decoder on two fp16 pieces (kind 4 of WPackJob): offsets into wsp and the chunks' powers of two; have_ench: built
    size_t enh_ch[4] = {0, 0, 0, 0}, eeh_ch[MAX_EDGE_SETS][2] = {}, deh_ch[2] = {0, 0};
    float enh_s[4] = {1, 1, 1, 1}, eeh_s[MAX_EDGE_SETS][2] = {}, deh_s[2] = {1, 1};
    bool have_ench = false;
    // static per-trajectory RHS inputs (mgn_set_static): cached encoded edge latents
    bool have_static = false;
    DevBuf stage;     // device staging image of caller-order latents (import / export)
    DevBuf lnall_v, lnall_e;   // mgn_processor_steps_dev under ln_dims = MGN_LN_ALL: the resident latents as caller-order rows for the unfused driver
    DevBuf gwork, gout, gpos, gtype;   // device-side graph prologue (csrc/graph_dev.hip): scratch, outputs, positions, node types
    DevBuf d_stamps;  // diagnostic builds only
    DevBuf ode;       // native rollout: state, stages, frames, saves, Elat0
    const float* srcA_override = nullptr;  // rollout: encoder reads the node state from here instead of d_nfA
    const float* elat_src_override = nullptr;   // right-hand sides on small meshes: step 0's edge kernel reads the trajectory's encoded edge latents from here (EdgeArgs::ElatSrc) instead of a restore copy into Elat
    float* out_override = nullptr;         // rollout: decoder writes dx/dt here instead of d_out
    DevBuf V, d_nfA, d_nfB, d_out, d_mask, d_sum;
    int32_t in_wa = 0, in_wb = 0;
    bool have_mask = false;
    bool lnall_edges = false;        // ln_dims = MGN_LN_ALL: the encoded edges of the resident static inputs sit in the whole-array arena

    // hipGraph of one mgn_processor_steps_dev(nsteps) pass: small meshes are launch-bound (3 kernels per step).
    // State machine per invalidation: first call runs eagerly (warms per-kernel attributes), second captures.
    int32_t use_graph = 1;          // MGN_GRAPH=0 disables
    int32_t graph_nsteps = -1;      // nsteps the cached graph was captured for
    int32_t graph_warm = -1;        // nsteps of the last eager run since the last invalidation
    hipGraphExec_t graph_exec = nullptr;
    // hipGraph of the resident right-hand side (mgn_ode_step after mgn_set_static) on small meshes; same life cycle
    hipGraphExec_t rhs_exec = nullptr;
    bool rhs_warm = false;
    // the same for mgn_forward (its device buffers keep their addresses between calls)
    hipGraphExec_t fwd_exec = nullptr;
    bool fwd_warm = false;

    // communicator (mgn_comm_init): the halo exchange and the staged schedule run inside the library
    mgn::Comm* comm = nullptr;
    int32_t force_staged = 0;                 // MGN_FORCE_STAGED=1: staged schedule even at nranks == 1 (self-test)
    DevBuf halo_send, halo_recv, gath_s, gath_r;
    std::vector<size_t> hx_sb, hx_so, hx_rb, hx_ro;   // per-peer byte counts / offsets of one exchange (rebuilt per graph)
    bool hx_ready = false;
    bool hx_direct = false;                   // rows are received straight into the halo block of P (one edge set)
    std::vector<int32_t> all_gid;             // [nranks][1 + max_own]: every rank's (n_own, own_gid...) for output gathers
    int32_t max_own = 0;
    bool in_local = false;                    // d_nfA / d_nfB / d_ef hold LOCAL rows (nranks > 1 uploads only what it owns)

    // training step (mgn_step): weights in training order, kept activations, scratch -- created on first use
    mgn::TrainState* train = nullptr;

    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> event_pool;   // recycled by mgn_profile_read: creating events inside the timed region costs microseconds each
};


namespace mgn {

int fail(mgn_engine* h, int code, const char* fmt, ...);
int need(mgn_engine* h, bool params, bool graph, bool packed = true, bool lnall_ok = false);   // packed = false: the caller reads h->params only (training, get_params)
int pack_inference_weights(mgn_engine* h);
// L x L chunk of W (row-major [K][ldw], rows kbase.., all L output columns) -> MFMA fragment order
void pack_chunk(float* dst, const float* W, int ldw, int kbase, int L);
void pack_chunk_tmajor(float* dst, const float* frag, int L);
void pack_chunk16(float* dst, const float* W, int ldw, int kbase);
// vector of L values (stride between consecutive features = stride) -> table fragment order
void pack_tab(float* dst, const float* vec, int L, int stride = 1);
// mgn_train.cpp: drop training-side state that depends on the parameters (what & 1) or the graph (what & 2); free it all
void train_invalidate(mgn_engine* h, int what);
void train_free(mgn_engine* h);
// mgn_config.ln_dims = MGN_LN_ALL (whole-array LayerNorm): the unfused forward (mgn_train.cpp)
int lnall_forward(mgn_engine* h, const float* nf, const float* ef, float* out);
int lnall_processor_steps(mgn_engine* h, float* v, float* e, int32_t nsteps);
int lnall_rhs_prepare(mgn_engine* h);                                              // binds the arena (may allocate / copy: outside of any capture)
int lnall_rhs_dev(mgn_engine* h, const float* srcA, float* out, bool reuse_edges);  // the right-hand side on resident inputs; launches only

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return mgn::fail(h, _e == hipErrorOutOfMemory ? MGN_E_OOM : MGN_E_HIP, "%s failed: %s (%s:%d)", #expr, \
                             hipGetErrorString(_e), __FILE__, __LINE__);                             \
    } while (0)

}  // namespace mgn

// No C++ exception crosses the C ABI (a Julia or C host would see std::terminate): every entry point is a function-try-block.
#define MGN_CATCH(h)                                                                                              \
    catch (const std::bad_alloc&) { return mgn::fail(h, MGN_E_OOM, "%s: host allocation failed", __func__); }     \
    catch (const std::exception& mgn_ex_) { return mgn::fail(h, MGN_E_ARG, "%s: %s", __func__, mgn_ex_.what()); }             \
    catch (...) { return mgn::fail(h, MGN_E_ARG, "%s: unknown C++ exception", __func__); }
#define MGN_CATCH_SIZE catch (...) { return 0; }
