// Kernel argument blocks and launch wrappers shared by kernels.hip and mgn_api.cpp.
// Internal to the engine; the public boundary is include/mgn_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgn {

constexpr int TILE = 32;        // rows (edges or nodes) per wave tile == MFMA 32x32 N dimension
constexpr int MAX_CHUNKS = 9;   // weight chunks (L x L, fragment order) a fused kernel may chain

// Table slots (each L floats, fragment order) inside a kernel's `tabs` block.
// T_LN: the LayerNorm variant as two numbers (eps_in, eps_out): rstd = 1 / (sqrt(var + eps_in) + eps_out)  (frag.hpp: ln_rstd)
enum { T_B1 = 0, T_B2, T_B3, T_GAMMA, T_BETA, T_BQ, T_LN, T_COUNT };

// General hidden-layer count (reference Args.hidden_layers, src/MeshGraphNets.jl:35-38; MGN-spec: h hidden layers = h + 1 Dense).
// The Dense layers AFTER the first one of an MLP: chunk[0 .. nmid-1] = the L x L middle layers (each followed by ReLU),
// chunk[nmid] = the last layer (absent for the decoder, whose last layer is L -> O on the VALU); tabs = their bias tables, L floats
// each, fragment order.  hidden_layers = 2 is nmid = 1.  Used by the GEN instantiations of the kernels (weights streamed from L2);
// the tuned kernel families are specialised for nmid = 1.
struct GenMlp {
    const float* chunk[4];
    const float* tabs;
    int32_t nmid;
    int32_t use;            // 1: launch the GEN instantiation
};

struct EdgeArgs {
    const int32_t* snd;     // [E] local sender index (may point into halo rows of P)
    const int32_t* rcv;     // [E] local receiver index, non-decreasing
    int64_t E;
    int32_t ntiles;
    const float* P;         // [n_own+n_halo][L]  v * W1[0:L]      (sender part of edge-MLP layer 1)
    const float* Q;         // [n_own][L]         v * W1[L:2L]+b1  (receiver part + bias)
    float* Elat;            // tile-major [ntiles][16 pieces][64 lanes][4]: edge latents, updated in place
    const float* ElatSrc;   // (k_edge_coop16m only; launch_edge_step refuses it elsewhere) non-null: the e rows are READ from this array of the same layout and written to Elat -- step 0 of a right-hand side reads the trajectory's encoded edges where they are kept, no restore copy
    float* AGG;             // tile-major over NODE tiles: per-receiver sums of e'
    float* CARRY;           // row-major [2*ntiles+1][L]: partial sums of receiver runs that straddle edge tiles; last row = 0
    const float* chunk[MAX_CHUNKS];  // 0:W2 1:W3 2:W1[2L:3L]   (fragment order)
    const float* chunk_t[MAX_CHUNKS]; // same chunks in t-major fragment order (cooperative small-graph kernels); null if absent
    const float* tabs;      // T_COUNT * L floats (fragment order): b2,b3 in T_B2,T_B3; LN in T_GAMMA,T_BETA
    int32_t stagger;        // s_sleep(127) units by which waves 4..7 of a block start late
    int32_t tile0;          // the launch covers edge tiles [tile0, tile0 + ntiles) (interior / boundary split)
    unsigned long long* stamps;  // diagnostic builds only (MGN_DIAG_STAMPS), else null
    GenMlp gen;
    int32_t c16;            // 1: small graph, 16-row cooperative tiles on v_mfma_f32_16x16x4_f32 (both kernels of a step must agree:
                            //    the carry rows are then per 16-edge tile)
    int32_t bf;             // 16-row kernels in bf16 mode: P, Q, Elat, AGG, CARRY are bf16 arrays in the bf16 kernels' layouts (passed
                            //    through the float pointers); weights, tables and arithmetic stay fp32
    // split path (split.hip; MGN_FP32_SPLIT / mgn_debug_fp32_split): the three layers on the bf16 matrix cores with fp32 accuracy --
    // every fp32 operand split into three bf16 pieces, six piece products kept.  split[i]: chunk i as 3 x 16384 bf16 (hi, mid, lo
    // pieces, the bf16 kernels' fragment order); null: not available
    const uint16_t* split[3];
    const uint16_t* split16[3];   // the same pieces in the fragment order of v_mfma_f32_16x16x32_bf16 (k_edge_coop16m on the split path, fp32 and bf16 storage; mgn_api.cpp: pack_chunk16_bf16)
    // two-piece fp16 form of the same chunks (k_edge_ring_h; split_common.hpp): chunk i times the power of two h2_s[i] as 2 x 16384 fp16
    // (hi, lo; the bf16 pieces' fragment order), h2_rs[i] = 1 / h2_s[i]; h2_b2pos = max(0, max_k b2[k]); null: not available
    const uint16_t* splith[3];
    const uint16_t* split16h[3];  // ... and in the 16x16x32 fragment order (k_edge_coop16m<.., 2>), the same scales
    float h2_s[3], h2_rs[3], h2_b2pos;
};

struct NodeArgs {
    int32_t n;              // owned nodes
    int32_t ntiles;
    const int32_t* rowptr;  // [n+1] CSR by receiver over the local (receiver-sorted) edge list
    float* V;               // tile-major node latents, updated in place
    const float* AGG;
    const float* CARRY;
    float* P;
    float* Q;
    const float* chunk[MAX_CHUNKS];  // 0:W2 1:W3 2:W1[0:L] 3:W1[L:2L] 4:WP(next) 5:WQ(next) 6:W1[2L:3L] (second edge set)
    const float* chunk_t[MAX_CHUNKS];
    // second edge set (world edges): its aggregate is one more layer-1 input block; AGG2 == null with one set
    const int32_t* rowptr2; const float* AGG2; const float* CARRY2; int64_t zero_row2;
    const float* tabs;      // b1,b2,b3,gamma,beta,bq
    int32_t mode;           // 0: MLP only (last step)  1: MLP + project P,Q  2: project only
    int32_t stagger;
    int64_t zero_row;       // CARRY row that is all zeros (read for receivers without incoming edges)
    int32_t tile0;          // k_project only: first tile of the range [tile0, tile0 + ntiles)
    GenMlp gen;
    int32_t c16;            // see EdgeArgs
    // 16-row kernels with two edge sets, mode 1: the second set's projection in the same launch (chunk[7] = WP, chunk[8] = WQ of
    // set 1, tabs2 = its tables: T_BQ holds b1 of set 1's edge MLP)
    float* P2; float* Q2; const float* tabs2;
    int32_t bf;             // see EdgeArgs (V, AGG, CARRY, P, Q bf16)
    unsigned long long* stamps;  // diagnostic builds only (MGN_DIAG_STAMPS), else null
    // split path (see EdgeArgs.split): chunk[0..5] as 3 x 16384 bf16 pieces each (k_node_split, k_project_split); null: not available
    const uint16_t* split[7];     // [6]: W1[2L:3L], the second edge set's aggregate block (k_node_split<true>)
    const uint16_t* split16[9];   // chunk[0..8] as pieces in the 16x16x32 fragment order (k_node_coop16 on the split path); null: not available
    // chunk[0..5] as two fp16 pieces times h2_s[i] (k_node_split_h, k_project_split_h; see EdgeArgs.splith); h2_b2pos = max(0, max_k b2[k])
    const uint16_t* splith[6];
    const uint16_t* split16h[9];  // chunk[0..8] in the 16x16x32 fragment order (k_node_coop16<.., 2>), scales h2_s[0..8]
    float h2_s[9], h2_rs[9], h2_b2pos;
};

struct EncNodeArgs {
    int32_t n, ntiles;
    const int32_t* gid;     // [n] global node id of each owned node (row into srcA/srcB)
    const float* srcA; int32_t wa;  // [N][wa]  first wa input features
    const float* srcB; int32_t wb;  // [N][wb]  remaining features (may be null, wb = 0)
    const float* scale;     // [Fn] input affine (normaliser), may be null
    const float* shift;
    const float* w1f;       // [Fn][L] first-layer weights, fragment order per input feature
    float* V; float* P; float* Q;
    const float* chunk[MAX_CHUNKS];  // 0:W2 1:W3 2:WP 3:WQ
    const float* tabs;
    GenMlp gen;
    const uint16_t* splith[4];       // the same chunks as two fp16 pieces (split_common.hpp), h2_rs[i] = 1 / chunk i's power of two; null: not built
    float h2_rs[4];
};

struct EncEdgeArgs {
    int64_t E; int32_t ntiles;
    const int64_t* gid;     // [E] global edge id of each local edge (row into ef)
    const float* ef; int32_t Fe;
    const float* scale; const float* shift;
    const float* w1f;
    float* Elat;
    const float* chunk[MAX_CHUNKS];  // 0:W2 1:W3
    const float* tabs;
    GenMlp gen;
    const uint16_t* splith[2];       // see EncNodeArgs
    float h2_rs[2];
};

struct DecArgs {
    int32_t n, ntiles;
    const float* V;
    const float* w3f;       // [O][L] last-layer weights, fragment order per output
    const float* b3;        // [O]
    int32_t O;
    const float* oscale; const float* oshift;  // [O] inverse normaliser, may be null
    const float* mask;      // [N] global val_mask, may be null
    const int32_t* gid;
    float* out;             // [n][O] local order
    const float* chunk[MAX_CHUNKS];  // 0:W1 1:W2
    const float* tabs;      // b1,b2
    GenMlp gen;             // decoder: chunk[0 .. nmid-1] = W2 .. W_h, no last chunk (L -> O runs on the VALU)
    const uint16_t* splith[2];       // see EncNodeArgs
    float h2_rs[2];
};

// ---- bf16 processor (BASELINE cfg-3 precision): bf16 storage + bf16 MFMA, fp32 accumulate / LayerNorm / residual /
// aggregation.  All pointers are bf16 (uint16_t) arrays; chunks are bf16 fragment order [s(8)][t(4)][lane(64)][8].
struct BfEdgeArgs {
    const int32_t* snd; const int32_t* rcv; int64_t E; int32_t ntiles; int32_t tile0;
    const uint16_t* P; const uint16_t* Q;      // row-major [row][16 pieces][8], fragment feature order
    uint16_t* Elat; uint16_t* AGG;             // tile-major [tile][8][64][8]
    uint16_t* CARRY;                           // row-major
    const uint16_t* chunk[3];                  // 0:W2 1:W3 2:W1e
    const float* tabs;                         // fp32 tables (same as the fp32 kernels)
    unsigned long long* stamps;                // diagnostic builds only (MGN_DIAG_STAMPS), else null
};
struct BfNodeArgs {
    int32_t n, ntiles; const int32_t* rowptr;
    uint16_t* V; const uint16_t* AGG; const uint16_t* CARRY; uint16_t* P; uint16_t* Q;
    const uint16_t* chunk[7];                  // 0:W2 1:W3 2:W1v 3:W1a 4:WP 5:WQ 6:W1a of the second edge set
    const float* tabs;
    int64_t zero_row; int32_t tile0;
    const int32_t* rowptr2; const uint16_t* AGG2; const uint16_t* CARRY2; int64_t zero_row2;   // second edge set or null
};
hipError_t launch_edge_bf16(const BfEdgeArgs& a, hipStream_t s);
hipError_t launch_node_bf16(const BfNodeArgs& a, hipStream_t s);      // node MLP
hipError_t launch_project_bf16(const BfNodeArgs& a, hipStream_t s);   // P,Q projection
// layout converters between the fp32 and bf16 tile-major forms (encoder output / decoder input in bf16 mode)
hipError_t launch_tile_f32_to_bf16(const float* src, uint16_t* dst, int64_t ntiles, hipStream_t s);
hipError_t launch_tile_bf16_to_f32(const uint16_t* src, float* dst, int64_t ntiles, hipStream_t s);
hipError_t launch_scatter_prows(const float* src, int src_stride /* floats */, float* dst, int64_t row0, int64_t rows, int L, hipStream_t s);   // plain rows -> P rows
hipError_t launch_scatter_prows16(const uint16_t* src, int src_stride /* elements */, uint16_t* dst, int64_t row0, int64_t rows, hipStream_t s);
bool prows_blocked();        // P / Q / CARRY rows are stored in blocks of eight (frag.hpp: prow_ptr)
hipError_t launch_gather_rows16(const uint16_t* src, const int32_t* idx, uint16_t* dst, int64_t rows, int dst_stride /* elements */, hipStream_t s);

struct LaunchCfg { int blocks; int threads; size_t lds; };
hipError_t launch_node_split_h(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s);    // split.hip: node MLP / projection on two fp16 pieces
hipError_t launch_project_split_h(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s);
hipError_t launch_edge_ring_h(const EdgeArgs& a, const LaunchCfg& lc, hipStream_t s);    // split.hip: the ring kernel on two fp16 pieces, three products
size_t edge_ring_h_lds();
int edge_ring_h_streamed();     // 1: launch_edge_ring_h runs k_edge_ring_hs (family codes 16 / 17), 0: k_edge_ring_h (13 / 14)
hipError_t launch_edge_ring(const EdgeArgs& a, const LaunchCfg& lc, hipStream_t s);    // split.hip
hipError_t launch_node_split(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s);    // split.hip
hipError_t launch_project_split(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s); // split.hip

struct LinComb {            // sum_j c[j] * k[j]
    int n;
    float c[7];
    const float* k[7];
};
hipError_t launch_lincomb(float* out, const float* u, const LinComb& lc, float dt, int64_t n, hipStream_t s);
hipError_t launch_overwrite(float* x, const float* frame, const uint8_t* mask, int64_t N, int O, hipStream_t s);
int errnorm_partials();
hipError_t launch_errnorm(const float* u, const float* unew, const LinComb& lc, float dt, float atol, float rtol, int64_t n,
                          double* partial, hipStream_t s);

// L in {32,64,128}.  All return hipError_t of the launch.
bool launch_is_small(int ntiles);
bool node_split_size(int ntiles);   // the split-path node kernels take launches of this many node tiles (above two per CU)
bool launch_is_small_edge(int ntiles_e);   // the same rule for edge launches (<= 16 tiles per CU)
int coop16_enabled();
int set_c16_edge_tiles(int t);   // edge tiles per CU up to which a handle takes the 16-row kernels (0: 3, or 2 where k_edge_ring_hs is the next family); returns the old value
bool ring_hs_default();          // fp32 launches above the cooperative range would run k_edge_ring_hs (split path on two fp16 pieces, streamed pieces)
bool coop16_size(int ntiles_e, int ntiles_n, bool ring_hs = false);   // the launch wrappers' rule for the cooperative node kernels (<= 8 tiles per CU)
int last_edge_kernel();
int last_node_kernel();         // family of the last fp32 edge launch (kernels.hip: launch_edge_step)
int set_fp32_split(int on);     // debug/tests: 0 = fp32-MFMA kernels, 1 = split path (default); returns the old value
int fp32_split_enabled();
int set_split_f16(int on);     // 1 (default): the split path on two fp16 pieces / three products where built (k_edge_ring_h), 0: three bf16 pieces / six products; MGN_SPLIT_F16; returns the old value
int split_f16_enabled();
int set_c16_split(int on);      // 16-row cooperative kernels on the split path (where fp32_split is on): bit 0 edge kernel at >= 2 row tiles, bit 1 node kernel (default 3), bit 2 edge kernel at one row tile too; MGN_C16_SPLIT
int c16_split_enabled();
int set_c16_row_tiles(int rt);  // debug/tests: 16-edge tiles per block of the small-graph edge kernel (0: chosen by size); returns the old value
int set_kernel_path(int p);   // debug/tests: 0 auto, 1 resident, 2 streaming, 3 cooperative, 4 GEN (general hidden_layers) kernels; returns the old value
int get_kernel_path();
hipError_t launch_edge_step(int L, const EdgeArgs& a, hipStream_t s);
hipError_t launch_node_step(int L, const NodeArgs& a, hipStream_t s);
hipError_t launch_project(int L, const NodeArgs& a, hipStream_t s);   // mode-2 work with both chunks LDS-resident
hipError_t launch_node_project_fused(int L, const NodeArgs& a, hipStream_t s, bool* launched);   // k_node_ring_hs where it applies (else *launched = false)
hipError_t launch_node_ring_hs(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s);            // split.hip
int node_ring_hs_enabled();
hipError_t launch_enc_node(int L, const EncNodeArgs& a, hipStream_t s);
hipError_t launch_enc_edge(int L, const EncEdgeArgs& a, hipStream_t s);
hipError_t launch_decode(int L, const DecArgs& a, hipStream_t s);
hipError_t launch_gather_rows(const float* src, const int32_t* idx, float* dst, int64_t rows, int L, int dst_stride /* elements */, hipStream_t s);
hipError_t launch_rows_to_tiles(const float* src, const int64_t* gid64, const int32_t* gid32, float* dst, int64_t rows, int L, hipStream_t s);
hipError_t launch_tiles_to_rows(const float* src, const int64_t* gid64, const int32_t* gid32, float* dst, int64_t rows, int L, hipStream_t s);
hipError_t launch_randn_rows(float* dst, const int64_t* gid64, const int32_t* gid32, int64_t rows, int L,
                             uint64_t seed, hipStream_t s);
// the MGN_PROW_BLOCK kernels.hip / split.hip were compiled with (mgn_create compares them with its own: frag.hpp prow_ptr)
int kernels_prow_block();
int split_prow_block();
int checksum_partials();  // doubles written by launch_checksum: (sum, sumsq) per block, to be added in order
hipError_t launch_checksum(const float* src, int64_t n, double* partials, hipStream_t s);

}  // namespace mgn
