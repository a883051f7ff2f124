// Kernel argument blocks and launch wrappers of the training step (mgn_step == GraphNetCore.step!, reference
// src/strategies.jl:418-422).  Internal; the public boundary is include/mgn_hip.h.
//
// All training tensors are ROW-MAJOR [rows][L] fp32 (rows padded to whole 32-row tiles by the allocator); inputs
// narrower than L (node / edge features, decoder output) are zero-padded to L so that every MLP of the model --
// encoders, processor MLPs, decoder -- runs through the same three MFMA kernels:
//   k_mlp_fwd   forward, keeps the post-ReLU hidden activations H1, H2 and the pre-LayerNorm output Y
//   k_mlp_bwd   LayerNorm / ReLU / Dense backward w.r.t. the activations (transposed weight chunks)
//   k_wgrad     dW = X^T G  (reduction over rows on the MFMA, deterministic per-block partials) + column sums of G
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgn {

struct TrainFwdArgs {
    int64_t rows; int32_t ntiles;
    const float* X[3]; const int32_t* xidx[3];   // layer-1 input blocks; xidx[j] != null: row gather
    const float* W1[3]; const float* W2; const float* W3;   // L x L chunks, fragment order (streamed from L2)
    const float* tabs;                           // T_B1, T_B2, T_B3, T_GAMMA, T_BETA (fragment order)
    float* H1; float* H2; float* Y;              // kept for the backward (null: not stored -- first pass of recompute mode)
    const float* resid; float* OUT;              // OUT = (resid ? resid : 0) + (ln ? LayerNorm(Y) : Y)
    float* LNOUT;                                // optional: LayerNorm(Y) alone (the message e' that is aggregated)
    // round 6, streaming kernels (train_fwd_fused_agg): the aggregation of e' = LayerNorm(Y) over runs of equal receiver inside the launch instead
    // of LNOUT + a segmented-sum pass -- rows are receiver-sorted, a run that lies inside a 32-row tile goes to SEG_AGG[receiver] (plain rows), the
    // pieces of a run that crosses tile borders to SEG_CARRY rows 2 tile (the part that continues from the tile before) / 2 tile + 1 (the part
    // that continues into the next); launch_seg_fixup adds the pieces up and zeroes the nodes without edges.  SEG_RCV: the receiver of every row.
    const int32_t* SEG_RCV; float* SEG_AGG; float* SEG_CARRY;
    int32_t ln;
    // factored first layer (edge MLP on large meshes): layer-1 pre-activation += PRE[i][preidx[i] ? preidx[i][row] : row]
    // (P = v W1_sender and Q = v W1_receiver, computed once per NODE by launch_lin2)
    const float* PRE[2]; const int32_t* preidx[2];
    // whole-array LayerNorm (ln = 0 here): (sum, sum of squares) of this launch's valid part of Y per slot, in double -- slot = tile for the
    // streaming kernels, 4 tile + wave for the cooperative ones (train_fwd_stat_slots); launch_array_stats_final adds the slots in order
    double* STATS;
};

struct TrainBwdArgs {
    int64_t rows; int32_t ntiles;
    const float* G0; const float* G1; const int32_t* g1idx;   // upstream gradient = G0[row] (+ G1[g1idx[row]])
    const float* Y; const float* H2; const float* H1;
    const float* W3T; const float* W2T; const float* W1T[3];  // transposed chunks, fragment order; W1T[j] null: skip
    const float* tabs;                           // gamma in T_GAMMA
    int32_t ln;                                  // 1: row-wise LayerNorm pullback here; 2: the whole-array one -- gy = rden (gamma g - m1 - xhat m2) with
    const float* LNS; const float* LNM;          //    LNS = (mean, rden, kappa) of the forward and LNM = (m1, m2) of launch_lnall_bwd; 0: none
    float* GT; float* GXH;                       // ln: total upstream gradient and G * xhat (-> dbeta, dgamma by column sums)
    // streaming kernels, 8 tiles per block (train_bwd_fused_sgr): the receiver-side segmented sum of GZ1 inside the launch, as the forward does it for e'
    // (TrainFwdArgs::SEG_*): runs inside a tile to SEG_OUT[receiver], pieces of runs that cross tile borders to SEG_CARRY, launch_seg_fixup after it
    const int32_t* SEG_RCV; float* SEG_OUT; float* SEG_CARRY;
    float* LNSUM;                                // ln == 1, non-null (streaming kernels, 8 tiles per block; train_bwd_ln_sums): [block][2][L] column sums of the block's rows of g (dbeta) and g * xhat (dgamma), reduced inside the kernel (DPP adds over a tile's 32 rows, the block's waves through LDS) -- neither GT / GXH nor LNROW is written then
    float* LNROW;                                // ln == 1, non-null: (mean, 1 / denominator) of every row instead of GT / GXH -- the LayerNorm job of the weight-gradient launch rebuilds both from G0 (+ G1) and Y (round 6: - 6 GB written per edge MLP on M-1M)
    float* GY; float* GZ2; float* GZ1;           // gradients at the three Dense outputs (-> weight / bias gradients)
    float* GX[3]; const float* GXadd[3];         // GX[j][row] = (GXadd[j] ? GXadd[j][row] : 0) + GZ1[row] * W1T[j]
};

hipError_t launch_mlp_fwd(int L, int nin, const TrainFwdArgs& a, hipStream_t s);
bool train_fwd_fused_agg(int L, int ntiles);   // the forward launch of this size takes SEG_* (streaming kernels at L = 128; MGN_TRAIN_FUSED_AGG = 0: never)
// agg[n] = 0 for a node without edges, the sum of its carry rows for a node whose run of edges crosses tile borders (others: written by the forward kernel)
hipError_t launch_seg_fixup(int L, const int32_t* rowptr, const float* carry, float* agg, int32_t n, hipStream_t s);
hipError_t launch_mlp_bwd(int L, int nin, const TrainBwdArgs& a, hipStream_t s);
bool train_uses_coop(int L, int ntiles);
int set_train_f16(int on);                 // 1 (default): streaming training kernels at L = 128 on two fp16 pieces / three products, 0: fp32 MFMA; returns the old value     // the cooperative 4-wave MLP kernels serve this launch size

// Two L x L products per row tile (the per-node halves of the factored first edge layer):
//   split  (X1 == null):  OUT0 = X0 W0,  OUT1 = X0 W1                     (P, Q of the forward)
//   merge  (X1 != null):  OUT0 = (ADD ? ADD : 0) + X0 W0 + X1 W1          (gradient w.r.t. v from the summed GZ1 rows)
struct Lin2Args {
    int64_t rows; int32_t ntiles;
    const float* X0; const float* X1;
    const float* W0; const float* W1;            // chunks in fragment order
    const float* ADD; float* OUT0; float* OUT1;
};
hipError_t launch_lin2(int L, const Lin2Args& a, hipStream_t s);
// out_r[n] = sum over the receiver range of src (rows contiguous), out_s[n] = sum over the sender range of src[perm[p]]
hipError_t launch_segment_sum_pair(int L, const float* src, const int32_t* rowptr_r, const int32_t* rowptr_s, const int32_t* perm_s,
                                   float* out_r, float* out_s, int32_t n, hipStream_t s);

// Weight gradients of one MLP in ONE launch (blockIdx.y = job):
//   dW[in][out] = sum_rows X[xidx ? xidx[row] : row][in] * G[row][out];   db[out] = sum_rows G[row][out]
// The row range is split over wgrad_blocks(rows) blocks: pw [nblocks][L][L], pb [nblocks][L] hold per-block partials.
// pw == null: column sums only (LayerNorm parameter gradients); pb == null: no column sums.
constexpr int WGRAD_MAX_JOBS = 32;   // (a launch may carry the jobs of several launch units: mgn_train.cpp, weight-gradient groups)
struct WgradJob {
    const float* X; const int32_t* xidx; const float* G; int64_t rows; float* pw; float* pb;
    // LayerNorm job (Y != null; X == null, pw == null; k_wgrad_h2 only -- wgrad_ln_jobs()): g = G[row] + (G1 ? G1[g1idx ? g1idx[row] : row] : 0),
    // xhat = (Y[row] - LNROW[2 row]) * LNROW[2 row + 1];  pb <- column sums of g (dbeta), pb2 <- column sums of g * xhat (dgamma)
    const float* Y; const float* LNROW; const float* G1; const int32_t* g1idx; float* pb2;
};
bool train_bwd_fused_sgr(int L, int ntiles);  // the backward launch of this size takes SEG_* (MGN_TRAIN_FUSED_SGR)
bool train_bwd_ln_sums(int L, int ntiles);   // the backward launch of this size takes LNSUM (MGN_TRAIN_BWD_LN_SUMS, default in train.hip)
// out[g][c] = sum of part[b][c] over the g-th of `groups` equal ranges of the nblocks blocks, in order (c < cols): the first level of the LNSUM reduction
hipError_t launch_colsum_groups(const float* part, int nblocks, int cols, int groups, float* out, hipStream_t s);
bool wgrad_ln_jobs(int L);                   // the weight-gradient launch at this L takes LayerNorm jobs
struct WgradBatch { int32_t njobs; int64_t rows_per_block; WgradJob job[WGRAD_MAX_JOBS]; };
int wgrad_blocks(int64_t rows);
int wgrad_blocks_of_job(int64_t launch_rows, int64_t job_rows);   // blocks of a launch sized for launch_rows that touch a job with fewer rows
// one L x L chunk of the inference layouts, from rows [kbase, kbase + L) of the matrix at params + src (leading dimension ldw).
// kind 0: fp32, three copies at wfrag + off (fragment order, t-major at + L L, 16x16x4 order at + 2 L L when L = 128);
// kind 1 / 2: the three exact bf16 pieces (hi, mid, lo; 16 384 each) at wsp + off in the 32x32x16 / 16x16x32 fragment order;
// kind 3: one bf16 copy at wbf + off (bf16 storage mode);
// kind 4: the two fp16 pieces (hi, lo; 16 384 each) of the chunk times `scale` (a power of two) at wsp + off, 32x32x16 fragment order
// (split_common.hpp); kind 5: the same in the 16x16x32 fragment order.  The host functions of mgn_api.cpp with the same names are the specification.
struct WPackJob { int kind; long long off, src; int ldw, kbase; float scale; };
hipError_t launch_pack_weights(int L, const WPackJob* jobs, int njobs, const float* params, float* wfrag, uint16_t* wsp, uint16_t* wbf, hipStream_t s);
// one packed copy of the training weights: kind 0 = an L x L chunk in fragment order + its t-major copy at + L * L, from rows [r0, r0 + nr)
// x cols [0, nc) of the matrix at params + src (leading dimension ldw; src < 0: the identity), zero-padded, transposed on request;
// kind 1 = T_COUNT * L table floats copied from tabs + src
struct PackJob { long long off, src; int ldw, r0, nr, nc, transpose, kind; float scale; };   // scale > 0 (L = 128): + the chunk's two fp16 pieces at off + 2 L L (32x32x16 fragment order: hi, lo), times the power of two that puts its largest entry into [2^14, 2^15) (taken on the device), and the inverse of that power at off + 3 L L
hipError_t launch_pack_train(int L, const PackJob* jobs, int njobs, const float* params, const float* tabs, unsigned* jobmax /* [njobs] scratch */, float* out, hipStream_t s);
hipError_t launch_wgrad(int L, WgradBatch wb, int64_t rows, hipStream_t s);
// out[r * cols + c] = sum_b partial[b * block_stride + r * ld + c]   (fixed order: bitwise reproducible), one job per blockIdx.y
constexpr int REDUCE_MAX_JOBS = 64;
struct ReduceJob { const float* partial; int32_t nblocks; int64_t block_stride; int32_t nrows, cols, ld; float* out; };
struct ReduceBatch { int32_t njobs; ReduceJob job[REDUCE_MAX_JOBS]; };
hipError_t launch_reduce_partials(const ReduceBatch& rb, hipStream_t s);
// out[n] = (add ? add[n] : 0) + sum_{p in [rowptr[n], rowptr[n+1])} src[perm ? perm[p] : p]   (rows of L floats)
hipError_t launch_segment_sum(int L, const float* src, const int32_t* rowptr, const int32_t* perm, const float* add, float* out,
                              int32_t n, hipStream_t s);
// out[n] = add[n] + sum_{p in A-range(n)} srcA[p] + sum_{p in B-range(n)} srcB[permB[p]]   (add may alias out)
hipError_t launch_segment_sum2(int L, const float* srcA, const int32_t* rowptrA, const float* srcB, const int32_t* rowptrB, const int32_t* permB,
                               const float* add, float* out, int32_t n, hipStream_t s);
// dst [rows][L] = [ (srcA | srcB)[rows][wa + wb] * scale + shift | 0 ]   (srcB may be null with wb = 0; scale null: identity)
hipError_t launch_affine_pad(const float* srcA, int wa, const float* srcB, int wb, const float* scale, const float* shift, float* dst, int L,
                             int64_t rows, hipStream_t s);
// the same with a row gather: dst row r <- source row gid[r] (gid null: row r)
hipError_t launch_affine_pad_gather(const float* srcA, int wa, const float* srcB, int wb, const float* scale, const float* shift, const int32_t* gid,
                                    float* dst, int L, int64_t rows, hipStream_t s);
// whole-array LayerNorm (mgn_config.ln_dims = MGN_LN_ALL): stats = (mean, 1 / (sqrt(var + eps_in) + eps_out), kappa) over the n values of x
// (double accumulation, fixed order; partial: 2 * array_stats_blocks() doubles), then t = (y - mean) * rden * gamma + beta
int array_stats_blocks();
int train_fwd_stat_slots(int L, int ntiles);     // slots a launch_mlp_fwd with TrainFwdArgs::STATS writes (2 doubles each)
hipError_t launch_array_stats_final(const double* partial, int64_t slots, int64_t n, float eps_in, float eps_out, float* stats, hipStream_t s);
hipError_t launch_array_stats(const float* x, int64_t n, double* partial, float eps_in, float eps_out, float* stats, hipStream_t s);
hipError_t launch_ln_all_apply(const float* y, const float* stats, const float* gamma, const float* beta, const float* resid, float* out,
                               float* lnout, int64_t n, int L, hipStream_t s);
// one pass over the receiver CSR: t = LN_all(y[p]); e[p] += t; agg[n] = sum of t over the edges node n receives (edge order)
hipError_t launch_ln_all_apply_segsum(const float* y, const float* stats, const float* gamma, const float* beta, float* e, const int32_t* rowptr,
                                      float* agg, int32_t n, int L, hipStream_t s);
// epilogue of a right-hand side: out [N][O] = (Y[:, 0:O] os + osh) .* mask[gid ? gid[row] : row]   (os / mask / gid may be null)
hipError_t launch_rhs_epilogue(const float* Y, int L, int O, const float* os, const float* osh, const float* mask, const int32_t* gid, float* out,
                               int64_t N, hipStream_t s);
// reverse pass of the whole-array LayerNorm of one MLP: dgamma, dbeta (L floats each, written into the gradient vector) and m = (m1, m2)
// (2 floats) for G = G0[row] (+ G1[g1idx ? g1idx[row] : row]); the MLP backward kernel then maps G to rden (gamma G - m1 - xhat m2) as it
// loads it (TrainBwdArgs::ln = 2).  partial: 2 * 128 * lnall_bwd_blocks() doubles; stats: (mean, rden, kappa) of the forward
int lnall_bwd_blocks();
hipError_t launch_lnall_bwd(const float* G0, const float* G1, const int32_t* g1idx, const float* Y, const float* stats, const float* gamma,
                            int64_t rows, int L, double* partial, float* m, float* dgamma, float* dbeta, hipStream_t s);
// node rows between the caller's order and the engine's (a renumbered graph: graph_host.h): gather dst[i] = src[gid[i]], scatter dst[gid[i]] = src[i]
hipError_t launch_permute_rows(float* dst, const float* src, const int32_t* gid, int64_t rows, int width, bool scatter, hipStream_t s);
// seed of the RHS VJP (mgn_ode_vjp): G[n][o] = lambda[n][o] * val_mask[n] * os[o]; optionally dxdt = (Y * os + osh) .* val_mask
hipError_t launch_vjp_seed(const float* Y, int L, int O, const float* lambda, const float* vm, const float* os, const float* osh, float* G,
                           float* dxdt, int64_t N, hipStream_t s);
// dst [N][O] = src[N][L][:, 0:O] .* scale
hipError_t launch_extract_cols(const float* src, int L, int O, const float* scale, float* dst, int64_t N, hipStream_t s);
// column statistics for the online normalisers: partial[b][0][f] = sum over block b's rows of x[r][f], partial[b][1][f] = sum of
// squares, in double, fixed order inside a block (stats_blocks(rows) blocks of STATS_ROWS rows; the caller adds the blocks in order)
constexpr int STATS_ROWS = 2048;
int stats_blocks(int64_t rows);
hipError_t launch_col_stats(const float* x, int64_t rows, int dim, double* partial, hipStream_t s);
// masked MSE: loss_partial[b] = sum over this block's mask entries of sum_o (out - target)^2;
// G[n][o] += 2 (out[n][o] - target[n][o]) / nmask   (G [N][L] zeroed by the caller; out = first O columns of Y)
int loss_blocks(int64_t nmask);
hipError_t launch_loss(const float* Y, int L, const float* target, int O, const int32_t* mask, int64_t nmask, int32_t index_base,
                       float* G, double* loss_partial, hipStream_t s);

}  // namespace mgn
