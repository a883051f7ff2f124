// CDNA4 (gfx950) kernels of the training step: forward with kept activations, backward, weight gradients.
//
// Replaces what Zygote + Lux + NNlib execute for GraphNetCore.step!(mgn, graph, target, mask, mse_reduce)
// (reference src/strategies.jl:418-422; gradients consumed at src/MeshGraphNets.jl:370-378).  First correct HIP
// version of SURVEY.md N2: the same lane-per-row MFMA building blocks as the inference kernels (frag.hpp), one
// MLP per launch, row-major activations, weights streamed from L2.  See train.h for the kernel inventory.
#include "train.h"

#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "frag.hpp"
#include "tile_common.hpp"
#include "split_common.hpp"

namespace mgn {

namespace {

struct RowRef {
    int64_t row, rr;
    bool valid;
};

DEVINL RowRef row_of(int tile, int c, int64_t rows) {
    RowRef r;
    r.row = (int64_t)tile * TILE + c;
    r.valid = r.row < rows;
    r.rr = r.valid ? r.row : rows - 1;
    return r;
}

}  // namespace

// the segmented sum of a fragment over runs of equal receiver inside a 32-row tile and its stores (k_mlp_bwd: GZ1 -> SGr): see k_mlp_fwd
template <int NT>
DEVINL void seg_sum_store(f32x16 (&acc)[NT], const int32_t* rcv, float* out, float* carry, int tile, int64_t rows, const RowRef& rw, int c, int h) {
    constexpr int L = 32 * NT;
    const int64_t e0 = (int64_t)tile * TILE;
    const int r = rcv[rw.rr];
    const int r_before = rcv[e0 > 0 ? e0 - 1 : 0], r_after = rcv[e0 + TILE < rows ? e0 + TILE : rows - 1];
    const int reff = rw.valid ? r : (-4 - c);
    const int rprev = __shfl_up(reff, 1, 32);
    const int rnext = __shfl_down(reff, 1, 32);
    const bool head = (c == 0) || (reff != rprev);
    const unsigned hm = (unsigned)__ballot(head);
    const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
    const int st_in = max(start, c & 16);
    const bool c1 = (c - 1 >= st_in), c2 = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
    const bool cx = (c >= 16) && (start <= 15);
    segmented_scan<NT>(acc, c1, c2, c4, c8, cx);
    const bool tail = rw.valid && ((c == 31) || (reff != rnext));
    const int r_first = __builtin_amdgcn_readfirstlane(reff);
    const bool sl = (start == 0) && e0 > 0 && (r_before == r_first);
    const bool sr = (c == 31) && (e0 + TILE < rows) && (r_after == reff);
    const bool to_carry = sl || sr;
    f32x4* dst = to_carry ? row_ptr(carry, (int64_t)2 * tile + (sl ? 0 : 1), L, h) : row_ptr(out, r, L, h);
    if (tail) store_frag<NT>(dst, STRIDE_ROW, acc);
}

// the kept activations are written once and read a whole pass later: stores that do not allocate in the caches (-DMGN_TRAIN_NT_STORES=1) were
// tried -- M-1M step 0.308 -> 0.319 s on the same box -- and are off
#ifndef MGN_TRAIN_NT_STORES
#define MGN_TRAIN_NT_STORES 0
#endif
template <int NT>
DEVINL void store_frag_keep(f32x4* __restrict__ p, int stride, const f32x16 (&x)[NT]) {
#if MGN_TRAIN_NT_STORES
#pragma unroll
    for (int m = 0; m < 4 * NT; ++m) {
        f32x4 v;
        v[0] = x[m >> 2][4 * (m & 3) + 0]; v[1] = x[m >> 2][4 * (m & 3) + 1];
        v[2] = x[m >> 2][4 * (m & 3) + 2]; v[3] = x[m >> 2][4 * (m & 3) + 3];
        __builtin_nontemporal_store(v, &p[m * stride]);
    }
#else
    store_frag<NT>(p, stride, x);
#endif
}

// ================================================================================================
// forward of one 3-Dense MLP (+ LayerNorm, + residual), keeping H1, H2, Y
// ================================================================================================
// Weight chunks are staged through a double-buffered LDS area by the whole block (4 waves = 4 tiles): chunk i+1 is
// copied while chunk i feeds the MFMA chain.  (Per-wave streaming from L2 through the 4-deep register ring of the
// persistent inference kernels is latency-bound here: one wave per SIMD, 5 chunks per tile.)
// Block-wide copy of one L x L chunk global -> LDS with all of a thread's loads in flight at once (copy_to_lds' runtime
// loop serialises load -> ds_write per iteration: ~11 us per 64 KiB chunk, which dominated these kernels).
template <int CH, int NTHR>
DEVINL void stage_chunk(float* dst, const float* __restrict__ src) {
    static_assert(CH / 4 >= NTHR || CH / 4 == 256, "chunk smaller than the block");
    constexpr int PER = CH / 4 / NTHR > 0 ? CH / 4 / NTHR : 1;   // float4 per thread: 16 (L = 128, 256 threads), 4 (L = 64), 1 (L = 32)
    constexpr int B = PER < 16 ? PER : 16;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    if (CH / 4 < NTHR && (int)threadIdx.x >= CH / 4) return;      // (L = 32 chunk under a 512-thread block: not instantiated)
#pragma unroll
    for (int base = 0; base < PER; base += B) {
        f32x4 v[B];
#pragma unroll
        for (int u = 0; u < B; ++u) v[u] = s4[(base + u) * NTHR + threadIdx.x];
#pragma unroll
        for (int u = 0; u < B; ++u) d4[(base + u) * NTHR + threadIdx.x] = v[u];
    }
}

// `cur` is a compile-time constant at every use (the chunk sequence is fully unrolled), so the LDS addresses are too.
#define CP_PRIME(first)                   \
    int cur = 0;                          \
    stage_chunk<CH, 64 * WPB>(smem, (first));       \
    __syncthreads()
#define CP_PREFETCH(next)                                                        \
    do {                                                                         \
        const float* nx_ = (next);                                               \
        if (H2) __builtin_amdgcn_sched_barrier(0);     /* (H2: the staging registers must not overlap the chain's 240) */ \
        if (nx_) stage_chunk<CH, 64 * WPB>(smem + (cur ^ 1) * CH, nx_);                    \
        if (H2) __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)
#define CP_W() (smem + cur * CH)
#define CP_ADVANCE()     \
    do {                 \
        __syncthreads(); \
        cur ^= 1;        \
    } while (0)

// One chunk of a chain: acc += in W.  H2 (round 5: L = 128, the streaming kernels of large launches): W staged as its two fp16 pieces
// (packed behind the two fp32 copies of every chunk, + 2 CH, times the power of two whose inverse sits at + 3 CH), the input split on
// the fly with its row scale, three piece products -- 96 v_mfma_f32_32x32x16_f16 of 32 cycles instead of 256 v_mfma_f32_32x32x2_f32 of
// 64 (split_common.hpp: h2_chunk_inplace).  Otherwise the fp32 MFMA chain on the fragment-order copy.
template <int NT, bool H2>
DEVINL void train_chunk(f32x16 (&acc)[NT], const f32x16 (&in)[NT], f32x16 (&part)[NT], const float* wlds, int lane, float rsw) {
    if constexpr (H2) {
        static_assert(NT == 4, "the fp16 pieces exist at L = 128");
        (void)part;
        h2_chunk_inplace(acc, in, wlds, lane, rsw);      // (in place: with a separate partial sum three 64-register arrays live across the chain and
                                                         //  the multi-block kernels spill under two waves per SIMD)
    } else {
        mfma_chunk<NT, true>(acc, in, wlds, lane);
    }
}
// where a chunk's staged data starts, and the inverse of its scale
#define HW(p) (H2 ? ((p) ? (p) + 2 * CH : nullptr) : (p))
#define HRS(p) (H2 ? (p)[3 * CH] : 1.f)

// WPB waves (= tiles) per block share the staged chunks: 4, or 8 on large launches (two waves per SIMD: one's loads and stores
// overlap the other's MFMA chain; 128 KiB of LDS allow one block per CU either way)
template <int NT, int NIN, int WPB, bool H2 = false>
__global__ __launch_bounds__(64 * WPB) void k_mlp_fwd(const TrainFwdArgs a) {
    constexpr int L = 32 * NT, CH = L * L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_raw = blockIdx.x * WPB + wave;
    const bool active = tile_raw < a.ntiles;
    const int tile = active ? tile_raw : a.ntiles - 1;       // idle waves shadow the last tile (no stores): they share the barriers
    OPAQUE_LANE();
    RowRef rw = row_of(tile, c, a.rows);
    rw.valid = rw.valid && active;
    CP_PRIME(HW(a.W1[0]));
    f32x16 x[NT], acc[NT], y[NT];
    tab_frag<NT>(acc, a.tabs + T_B1 * L, h);
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (a.PRE[i]) add_frag<NT>(acc, row_ptr(a.PRE[i], a.preidx[i] ? (int64_t)a.preidx[i][rw.rr] : rw.rr, L, h), STRIDE_ROW);
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
        CP_PREFETCH(HW(j + 1 < NIN ? a.W1[j + 1] : a.W2));
        const int64_t src = a.xidx[j] ? (int64_t)a.xidx[j][rw.rr] : rw.rr;
        load_frag<NT>(x, row_ptr(a.X[j], src, L, h), STRIDE_ROW);
        train_chunk<NT, H2>(acc, x, y, CP_W(), lane, HRS(a.W1[j]));
        CP_ADVANCE();
    }
    relu_frag<NT>(acc);
    if (a.H1 && rw.valid) store_frag_keep<NT>(row_ptr(a.H1, rw.row, L, h), STRIDE_ROW, acc);
    CP_PREFETCH(HW(a.W3));
    tab_frag<NT>(y, a.tabs + T_B2 * L, h);
    train_chunk<NT, H2>(y, acc, x, CP_W(), lane, HRS(a.W2));
    CP_ADVANCE();
    relu_frag<NT>(y);
    if (a.H2 && rw.valid) store_frag_keep<NT>(row_ptr(a.H2, rw.row, L, h), STRIDE_ROW, y);
    tab_frag<NT>(acc, a.tabs + T_B3 * L, h);
    train_chunk<NT, H2>(acc, y, x, CP_W(), lane, HRS(a.W3));
    if (a.Y && rw.valid) store_frag_keep<NT>(row_ptr(a.Y, rw.row, L, h), STRIDE_ROW, acc);
    if (a.STATS) {                               // whole-array LayerNorm: this tile's (sum, sum of squares) of Y, lanes added in a fixed pattern
        float s1 = 0.f, s2 = 0.f;
        if (rw.valid) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) { s1 += acc[t][k]; s2 += acc[t][k] * acc[t][k]; }
        }
        double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
        if (active && lane0 == 0) { a.STATS[2 * (int64_t)tile_raw] = d1; a.STATS[2 * (int64_t)tile_raw + 1] = d2; }
    }
    if (a.ln) layer_norm_frag<NT>(acc, a.tabs + T_GAMMA * L, a.tabs + T_BETA * L, h);
    if (a.LNOUT && rw.valid) store_frag<NT>(row_ptr(a.LNOUT, rw.row, L, h), STRIDE_ROW, acc);
    if constexpr (NT == 4 && WPB == 8) {
        if (a.SEG_RCV) {
            // OUT = resid + e' from a second array (x is free), then the segmented sum of e' over runs of equal receiver as in the inference
            // kernels (tile_common.hpp: segmented_scan; both halves of a row see the same structure)
            if (a.OUT) {
                if (a.resid) load_frag<NT>(x, row_ptr(a.resid, rw.rr, L, h), STRIDE_ROW);
                else zero_frag<NT>(x);
#pragma unroll
                for (int t = 0; t < NT; ++t) x[t] += acc[t];
                if (rw.valid) store_frag<NT>(row_ptr(a.OUT, rw.row, L, h), STRIDE_ROW, x);
            }
            seg_sum_store<NT>(acc, a.SEG_RCV, a.SEG_AGG, a.SEG_CARRY, tile, a.rows, rw, c, h);
            return;
        }
    }
    if (a.resid) add_frag<NT>(acc, row_ptr(a.resid, rw.rr, L, h), STRIDE_ROW);
    if (a.OUT && rw.valid) store_frag<NT>(row_ptr(a.OUT, rw.row, L, h), STRIDE_ROW, acc);
}

// the nodes the fused aggregation of k_mlp_fwd leaves open: no edges -> zeros; a run of edges over several tiles -> the sum of its carry rows
__global__ void k_seg_fixup(const int32_t* __restrict__ rowptr, const float* __restrict__ carry, float* __restrict__ agg, int32_t n, int L4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * L4) return;
    const int node = (int)(i / L4), q = (int)(i - (int64_t)node * L4);
    const int a0 = rowptr[node], a1 = rowptr[node + 1];
    if (a1 > a0) {
        const int T1 = a0 >> 5, T2 = (a1 - 1) >> 5;
        if (T2 == T1) return;
        const f32x4* C4 = reinterpret_cast<const f32x4*>(carry);
        f32x4 s = C4[(int64_t)(2 * T1 + 1) * L4 + q];
        for (int t = T1 + 1; t <= T2; ++t) s += C4[(int64_t)(2 * t) * L4 + q];
        reinterpret_cast<f32x4*>(agg)[i] = s;
    } else {
        reinterpret_cast<f32x4*>(agg)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ================================================================================================
// backward of the same MLP w.r.t. its activations
//   LayerNorm:  gy = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))
//   Dense i:    g_in = g_out * W_i^T (transposed chunk through the same MFMA path), masked by the kept ReLU output
// ================================================================================================
template <int NT>
DEVINL void mask_by_relu(f32x16 (&g)[NT], const f32x16 (&act)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) g[t][k] = act[t][k] > 0.f ? g[t][k] : 0.f;
}

// column sums of a fragment block over the tile's 32 rows (lanes c = 0..31 of each half): an inclusive DPP scan in place -- reach 1, 2, 4, 8
// inside the 16-lane rows, then lane 15 / 47 into the row above -- leaves the totals in lanes 31 and 63, which store them (feature of register
// k of block t in half h: 32 t + 8 (k >> 2) + 4 h + (k & 3)).  DPP reads of a register follow its last VALU write by >= 16 instructions.
DEVINL void colsum_block(f32x16& x, float* dst, int t, int c, int h) {
#define MGN_COLSUM_LEVEL(CTRL)                                                                         \
    _Pragma("unroll") for (int k = 0; k < 16; ++k)                                                     \
        asm volatile("v_add_f32_dpp %0, %0, %0 " CTRL " bound_ctrl:0" : "+v"(x[k]));
    asm volatile("s_nop 1");
    MGN_COLSUM_LEVEL("row_shr:1 row_mask:0xf bank_mask:0xf")
    MGN_COLSUM_LEVEL("row_shr:2 row_mask:0xf bank_mask:0xf")
    MGN_COLSUM_LEVEL("row_shr:4 row_mask:0xf bank_mask:0xf")
    MGN_COLSUM_LEVEL("row_shr:8 row_mask:0xf bank_mask:0xf")
    MGN_COLSUM_LEVEL("row_bcast:15 row_mask:0xa bank_mask:0xf")
#undef MGN_COLSUM_LEVEL
    if (c == 31) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(dst + 32 * t + 8 * g + 4 * h) = f32x4{x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
    }
}

template <int NT, int NIN, int WPB, bool H2 = false>
__global__ __launch_bounds__(64 * WPB) void k_mlp_bwd(const TrainBwdArgs a) {
    constexpr int L = 32 * NT, CH = L * L;
    constexpr float invL = 1.0f / L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_raw = blockIdx.x * WPB + wave;
    const bool active = tile_raw < a.ntiles;
    const int tile = active ? tile_raw : a.ntiles - 1;
    OPAQUE_LANE();
    RowRef rw = row_of(tile, c, a.rows);
    rw.valid = rw.valid && active;
    CP_PRIME(HW(a.W3T));
    f32x16 g[NT], y[NT], acc[NT];
    load_frag<NT>(g, row_ptr(a.G0, rw.rr, L, h), STRIDE_ROW);
    if (a.G1) add_frag<NT>(g, row_ptr(a.G1, a.g1idx ? (int64_t)a.g1idx[rw.rr] : rw.rr, L, h), STRIDE_ROW);
    if (!rw.valid) zero_frag<NT>(g);
    if (a.ln == 2) {                             // whole-array LayerNorm: gy = rden (gamma g - m1 - xhat m2), statistics and means given
        load_frag<NT>(y, row_ptr(a.Y, rw.rr, L, h), STRIDE_ROW);
        const float mean = a.LNS[0], rden = a.LNS[1], m1 = a.LNM[0], m2 = a.LNM[1];
        tab_frag<NT>(acc, a.tabs + T_GAMMA * L, h);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float xh = (y[t][k] - mean) * rden;
                g[t][k] = rw.valid ? rden * (acc[t][k] * g[t][k] - m1 - xh * m2) : 0.f;
            }
    }
    [[maybe_unused]] float* const lnacc = smem + 2 * CH + wave * (2 * L);      // (LNSUM) this wave's [dbeta | dgamma] sums
    if (a.ln == 1) {
        if (a.GT && rw.valid) store_frag<NT>(row_ptr(a.GT, rw.row, L, h), STRIDE_ROW, g);
        load_frag<NT>(y, row_ptr(a.Y, rw.rr, L, h), STRIDE_ROW);
        if constexpr (NT == 4 && WPB == 8) {
            if (a.LNSUM) {                       // dbeta: the column sums of g, one block at a time through acc[0] (free here) while Y is on its way
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[0] = g[t];
                    colsum_block(acc[0], lnacc, t, c, h);
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) s += y[t][k];
        s += __shfl_xor(s, 32, 64);
        const float mean = s * invL;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float d = y[t][k] - mean;
                y[t][k] = d;
                q += d * d;
            }
        q += __shfl_xor(q, 32, 64);
        // y = d / D, D = sqrt(var + eps_in) + eps_out (the T_LN slot: both LayerNorm variants, frag.hpp ln_rstd_at); dD / dx_i =
        // d_i / (L sqrt(var + eps_in)), so the xhat term of the pullback carries kappa = D / sqrt(var + eps_in) (1 when eps_out = 0)
        const float lsq = sqrtf(q * invL + a.tabs[T_LN * L]);
        const float rstd = 1.0f / (lsq + a.tabs[T_LN * L + 1]);
        const float kappa = lsq > 0.f ? (lsq + a.tabs[T_LN * L + 1]) / lsq : 1.f;
        if (a.LNROW && rw.valid && h == 0) *reinterpret_cast<float2*>(a.LNROW + 2 * rw.row) = float2{mean, rstd};
        tab_frag<NT>(acc, a.tabs + T_GAMMA * L, h);           // acc = gamma
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float xh = y[t][k] * rstd;
                const float gg = g[t][k] * acc[t][k];
                acc[t][k] = g[t][k] * xh;                     // G * xhat  (-> dgamma)
                y[t][k] = xh;
                g[t][k] = gg;
                m1 += gg;
                m2 += gg * xh;
            }
        m1 += __shfl_xor(m1, 32, 64);
        m2 += __shfl_xor(m2, 32, 64);
        m1 *= invL;
        m2 *= invL * kappa;
        if (a.GXH && rw.valid) store_frag<NT>(row_ptr(a.GXH, rw.row, L, h), STRIDE_ROW, acc);
        if constexpr (NT == 4 && WPB == 8) {
            if (a.LNSUM) {                       // dgamma: the column sums of g * xhat, in place (acc is dead from here)
#pragma unroll
                for (int t = 0; t < NT; ++t) colsum_block(acc[t], lnacc + L, t, c, h);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) g[t][k] = rstd * (g[t][k] - m1 - y[t][k] * m2);
    }
    if (rw.valid) store_frag<NT>(row_ptr(a.GY, rw.row, L, h), STRIDE_ROW, g);
    CP_PREFETCH(HW(a.W2T));
    zero_frag<NT>(acc);
    train_chunk<NT, H2>(acc, g, y, CP_W(), lane, HRS(a.W3T)); // gradient at H2
    CP_ADVANCE();
    load_frag<NT>(y, row_ptr(a.H2, rw.rr, L, h), STRIDE_ROW);
    mask_by_relu<NT>(acc, y);
    if (rw.valid) store_frag<NT>(row_ptr(a.GZ2, rw.row, L, h), STRIDE_ROW, acc);
    // the input-gradient chunks that exist, in order (kernel-uniform)
    const float* nxt[3] = {nullptr, nullptr, nullptr};
    int first = -1;
#pragma unroll
    for (int j = NIN - 1; j >= 0; --j)
        if (a.W1T[j]) { nxt[j] = first >= 0 ? a.W1T[first] : nullptr; first = j; }
    CP_PREFETCH(HW(first >= 0 ? a.W1T[first] : nullptr));
    zero_frag<NT>(g);
    train_chunk<NT, H2>(g, acc, y, CP_W(), lane, HRS(a.W2T)); // gradient at H1
    CP_ADVANCE();
    load_frag<NT>(y, row_ptr(a.H1, rw.rr, L, h), STRIDE_ROW);
    mask_by_relu<NT>(g, y);
    if (rw.valid) store_frag<NT>(row_ptr(a.GZ1, rw.row, L, h), STRIDE_ROW, g);
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
        if (!a.W1T[j]) continue;
        CP_PREFETCH(HW(nxt[j]));
        if (a.GXadd[j]) load_frag<NT>(acc, row_ptr(a.GXadd[j], rw.rr, L, h), STRIDE_ROW);
        else zero_frag<NT>(acc);
        train_chunk<NT, H2>(acc, g, y, CP_W(), lane, HRS(a.W1T[j]));
        CP_ADVANCE();
        if (rw.valid) store_frag<NT>(row_ptr(a.GX[j], rw.row, L, h), STRIDE_ROW, acc);
    }
    if constexpr (NT == 4 && WPB == 8) {
        if (a.SEG_RCV) seg_sum_store<NT>(g, a.SEG_RCV, a.SEG_OUT, a.SEG_CARRY, tile, a.rows, rw, c, h);   // g = GZ1, dead from here: SGr
        if (a.ln == 1 && a.LNSUM) {              // the block's eight waves, added in order
            __syncthreads();
            if (threadIdx.x < 2 * L) {
                const float* p = smem + 2 * CH + threadIdx.x;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < WPB; ++w) sum += p[w * 2 * L];
                a.LNSUM[(size_t)blockIdx.x * (2 * L) + threadIdx.x] = sum;
            }
        }
    }
}

// first level of the LNSUM reduction: group g of `groups` adds its range of blocks in order
__global__ void k_colsum_groups(const float* __restrict__ part, int nblocks, int cols, int groups, float* __restrict__ out) {
    const int per = (nblocks + groups - 1) / groups;
    const int b0 = blockIdx.x * per, b1 = min(nblocks, b0 + per);
    for (int cidx = threadIdx.x; cidx < cols; cidx += blockDim.x) {
        float s = 0.f;
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            const float v0 = part[(size_t)b * cols + cidx], v1 = part[(size_t)(b + 1) * cols + cidx];
            const float v2 = part[(size_t)(b + 2) * cols + cidx], v3 = part[(size_t)(b + 3) * cols + cidx];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < b1; ++b) s += part[(size_t)b * cols + cidx];
        out[(size_t)blockIdx.x * cols + cidx] = s;
    }
}

// ================================================================================================
// Cooperative 4-wave variants for SMALL meshes (L = 128): one 32-row tile per block, wave t owns feature block t of every
// Dense output (chains of 64 MFMAs instead of 256), the waves exchange their slices through LDS between layers and read
// their slice of every weight chunk from its t-major copy (at + L*L of the fragment-order chunk) through a register ring.
// A cylinder_flow-sized datapoint (374 edge tiles) then occupies ~1500 waves instead of 374 on 94 CUs.  Same arithmetic in
// the same order as k_mlp_fwd / k_mlp_bwd.
// ================================================================================================
DEVINL void add_quarter(f32x16& q, const f32x4* __restrict__ p, int stride, int t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = p[(4 * t + g) * stride];
        q[4 * g + 0] += v[0]; q[4 * g + 1] += v[1]; q[4 * g + 2] += v[2]; q[4 * g + 3] += v[3];
    }
}

// The chains of these kernels pin their weight ring with scheduling fences (coop_chain_primed<true>, frag.hpp): with many
// row-fragment registers live hipcc otherwise sinks each request to just before its use and every k-step group waits a full
// L2 round trip (3-4 x the chain's MFMA time).
// a chain of the cooperative kernels on the fp32 MFMA pipe (t-major copy of the chunk, frag.hpp) or, H2, on two fp16 pieces (the pieces
// packed behind the chunk's fp32 copies; split_common.hpp: h2c_chain_primed).  W: the chunk's base.
template <bool H2> struct TRing;
template <> struct TRing<false> { CoopRing r; };
template <> struct TRing<true> { H2CoopRing r; };
template <bool H2>
DEVINL void t_prime(TRing<H2>& g, const float* W, int tq, int lane) {
    if constexpr (H2) h2c_prime(g.r, h2c_w(W, tq, lane));
    else coop_prime(g.r, W + 128 * 128 + tq * 4096, lane);
}
template <bool H2>
DEVINL void t_chain_primed(f32x16& acc, const f32x16 (&in)[4], const float* W, int tq, int lane, TRing<H2>& g) {
    if constexpr (H2) h2c_chain_primed(acc, in, h2c_w(W, tq, lane), g.r, W[3 * 128 * 128]);
    else coop_chain_primed<true>(acc, in, W + 128 * 128 + tq * 4096, lane, g.r);
}
template <bool H2>
DEVINL void t_chain(f32x16& acc, const f32x16 (&in)[4], const float* W, int tq, int lane) {
    TRing<H2> g;
    t_prime<H2>(g, W, tq, lane);
    __builtin_amdgcn_sched_barrier(0);
    t_chain_primed<H2>(acc, in, W, tq, lane, g);
}

template <int NIN, bool H2 = false>
__global__ __launch_bounds__(256, 2) void k_mlp_fwd_coop(const TrainFwdArgs a) {
    constexpr int L = 128, CH = L * L, QS = 4096;     // QS: one wave's t-slice of a chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xch0 = reinterpret_cast<f32x4*>(smem);
    f32x4* xch1 = xch0 + 16 * 64;
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tq = wave;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        OPAQUE_LANE();
        const RowRef rw = row_of(tile, c, a.rows);
        int64_t src[NIN];
#pragma unroll
        for (int j = 0; j < NIN; ++j) src[j] = a.xidx[j] ? (int64_t)a.xidx[j][rw.rr] : rw.rr;
        f32x16 xa[4], xb[4], acc;
        TRing<H2> r2, r3;
        load_frag<4>(xa, row_ptr(a.X[0], src[0], L, h), STRIDE_ROW);
        if constexpr (NIN > 1) load_frag<4>(xb, row_ptr(a.X[1], src[1], L, h), STRIDE_ROW);
        tab_quarter(acc, a.tabs + T_B1 * L, tq, h);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (a.PRE[i]) add_quarter(acc, row_ptr(a.PRE[i], a.preidx[i] ? (int64_t)a.preidx[i][rw.rr] : rw.rr, L, h), STRIDE_ROW, tq);
        // Fences: without them hipcc sinks the (gathered, slow) row loads INTO the MFMA chains, a few pieces ahead of their use,
        // where they queue in front of the weight ring's loads (vmcnt retires in order) and every k-step waits for memory.
        PHASE_FENCE();
        t_chain<H2>(acc, xa, a.W1[0], tq, lane);
        PHASE_FENCE();
        if constexpr (NIN > 2) load_frag<4>(xa, row_ptr(a.X[2], src[2], L, h), STRIDE_ROW);
        if constexpr (NIN <= 2) t_prime<H2>(r2, a.W2, tq, lane);
        PHASE_FENCE();
        if constexpr (NIN > 1) t_chain<H2>(acc, xb, a.W1[1], tq, lane);
        PHASE_FENCE();
        if constexpr (NIN > 2) {
            t_prime<H2>(r2, a.W2, tq, lane);
            PHASE_FENCE();
            t_chain<H2>(acc, xa, a.W1[2], tq, lane);
        }
        relu_quarter(acc);
        if (a.H1 && rw.valid) store_quarter(row_ptr(a.H1, rw.row, L, h), STRIDE_ROW, tq, acc);
        coop_exchange(xa, acc, xch0, wave, lane);
        t_prime<H2>(r3, a.W3, tq, lane);
        tab_quarter(acc, a.tabs + T_B2 * L, tq, h);
        t_chain_primed<H2>(acc, xa, a.W2, tq, lane, r2);
        relu_quarter(acc);
        if (a.H2 && rw.valid) store_quarter(row_ptr(a.H2, rw.row, L, h), STRIDE_ROW, tq, acc);
        coop_exchange(xb, acc, xch1, wave, lane);
        tab_quarter(acc, a.tabs + T_B3 * L, tq, h);
        t_chain_primed<H2>(acc, xb, a.W3, tq, lane, r3);
        if (a.Y && rw.valid) store_quarter(row_ptr(a.Y, rw.row, L, h), STRIDE_ROW, tq, acc);
        if (a.STATS) {                                                 // whole-array LayerNorm: (sum, sum of squares) of this wave's quarter of Y
            float s1 = 0.f, s2 = 0.f;
            if (rw.valid) {
#pragma unroll
                for (int k = 0; k < 16; ++k) { s1 += acc[k]; s2 += acc[k] * acc[k]; }
            }
            double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
            if (lane0 == 0) { a.STATS[2 * ((int64_t)tile * 4 + wave)] = d1; a.STATS[2 * ((int64_t)tile * 4 + wave) + 1] = d2; }
        }
        if (a.ln) {
            coop_exchange(xa, acc, xch0, wave, lane);                  // full pre-LN row for the statistics
            coop_layer_norm(acc, xa, a.tabs + T_GAMMA * L, a.tabs + T_BETA * L, tq, h);
        }
        if (a.LNOUT && rw.valid) store_quarter(row_ptr(a.LNOUT, rw.row, L, h), STRIDE_ROW, tq, acc);
        if (a.resid) add_quarter(acc, row_ptr(a.resid, rw.rr, L, h), STRIDE_ROW, tq);
        if (a.OUT && rw.valid) store_quarter(row_ptr(a.OUT, rw.row, L, h), STRIDE_ROW, tq, acc);
        __syncthreads();                                               // the exchange buffers are rewritten by the next tile
    }
}

// feature block t (wave-uniform) of a full row fragment, without dynamic register indexing
DEVINL f32x16 pick_quarter(const f32x16 (&x)[4], int t) { return t == 0 ? x[0] : t == 1 ? x[1] : t == 2 ? x[2] : x[3]; }

DEVINL void mask_quarter(f32x16& g, const f32x16& act) {
#pragma unroll
    for (int k = 0; k < 16; ++k) g[k] = act[k] > 0.f ? g[k] : 0.f;
}

template <int NIN, bool H2 = false>
__global__ __launch_bounds__(256, 2) void k_mlp_bwd_coop(const TrainBwdArgs a) {
    constexpr int L = 128, CH = L * L, QS = 4096;
    constexpr float invL = 1.0f / L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xch0 = reinterpret_cast<f32x4*>(smem);
    f32x4* xch1 = xch0 + 16 * 64;
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tq = wave;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        OPAQUE_LANE();
        const RowRef rw = row_of(tile, c, a.rows);
        f32x16 g[4], y[4], acc, q;
        TRing<H2> r3, r2;
        t_prime<H2>(r3, a.W3T, tq, lane);
        // every wave holds the full upstream row (it is the B operand of the first chain) and repeats the row statistics
        load_frag<4>(g, row_ptr(a.G0, rw.rr, L, h), STRIDE_ROW);
        if (a.G1) add_frag<4>(g, row_ptr(a.G1, a.g1idx ? (int64_t)a.g1idx[rw.rr] : rw.rr, L, h), STRIDE_ROW);
        if (!rw.valid) zero_frag<4>(g);
        if (a.ln == 2) {                                               // whole-array LayerNorm (see k_mlp_bwd)
            load_frag<4>(y, row_ptr(a.Y, rw.rr, L, h), STRIDE_ROW);
            const float mean = a.LNS[0], rden = a.LNS[1], m1 = a.LNM[0], m2 = a.LNM[1];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                tab_quarter(q, a.tabs + T_GAMMA * L, t, h);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float xh = (y[t][k] - mean) * rden;
                    g[t][k] = rw.valid ? rden * (q[k] * g[t][k] - m1 - xh * m2) : 0.f;
                }
            }
        }
        if (a.ln == 1) {
            if (rw.valid) store_quarter(row_ptr(a.GT, rw.row, L, h), STRIDE_ROW, tq, pick_quarter(g, tq));
            load_frag<4>(y, row_ptr(a.Y, rw.rr, L, h), STRIDE_ROW);
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) s += y[t][k];
            s += __shfl_xor(s, 32, 64);
            const float mean = s * invL;
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float d = y[t][k] - mean;
                    y[t][k] = d;
                    v += d * d;
                }
            v += __shfl_xor(v, 32, 64);
            const float lsq = sqrtf(v * invL + a.tabs[T_LN * L]);              // (see k_mlp_bwd)
            const float rstd = 1.0f / (lsq + a.tabs[T_LN * L + 1]);
            const float kappa = lsq > 0.f ? (lsq + a.tabs[T_LN * L + 1]) / lsq : 1.f;
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                tab_quarter(q, a.tabs + T_GAMMA * L, t, h);            // gamma, feature block t
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float xh = y[t][k] * rstd;
                    const float gg = g[t][k] * q[k];
                    q[k] = g[t][k] * xh;                               // G * xhat  (-> dgamma)
                    y[t][k] = xh;
                    g[t][k] = gg;
                    m1 += gg;
                    m2 += gg * xh;
                }
                if (t == tq && rw.valid) store_quarter(row_ptr(a.GXH, rw.row, L, h), STRIDE_ROW, tq, q);
            }
            m1 += __shfl_xor(m1, 32, 64);
            m2 += __shfl_xor(m2, 32, 64);
            m1 *= invL;
            m2 *= invL * kappa;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) g[t][k] = rstd * (g[t][k] - m1 - y[t][k] * m2);
        }
        if (rw.valid) store_quarter(row_ptr(a.GY, rw.row, L, h), STRIDE_ROW, tq, pick_quarter(g, tq));
        t_prime<H2>(r2, a.W2T, tq, lane);
        load_quarter(q, row_ptr(a.H2, rw.rr, L, h), STRIDE_ROW, tq);
        PHASE_FENCE();                                                 // (see k_mlp_fwd_coop: loads stay outside the chains)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        t_chain_primed<H2>(acc, g, a.W3T, tq, lane, r3);     // gradient at H2
        mask_quarter(acc, q);
        if (rw.valid) store_quarter(row_ptr(a.GZ2, rw.row, L, h), STRIDE_ROW, tq, acc);
        coop_exchange(y, acc, xch0, wave, lane);
        load_quarter(q, row_ptr(a.H1, rw.rr, L, h), STRIDE_ROW, tq);
        PHASE_FENCE();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        t_chain_primed<H2>(acc, y, a.W2T, tq, lane, r2);     // gradient at H1
        mask_quarter(acc, q);
        if (rw.valid) store_quarter(row_ptr(a.GZ1, rw.row, L, h), STRIDE_ROW, tq, acc);
        coop_exchange(g, acc, xch1, wave, lane);
#pragma unroll
        for (int j = 0; j < NIN; ++j) {
            if (!a.W1T[j]) continue;
            if (a.GXadd[j]) load_quarter(acc, row_ptr(a.GXadd[j], rw.rr, L, h), STRIDE_ROW, tq);
            else {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            }
            t_chain<H2>(acc, g, a.W1T[j], tq, lane);
            if (rw.valid) store_quarter(row_ptr(a.GX[j], rw.row, L, h), STRIDE_ROW, tq, acc);
        }
        __syncthreads();
    }
}

// ================================================================================================
// two L x L products per row tile: the per-node halves of the factored first edge layer (see train.h, Lin2Args)
// ================================================================================================
template <int NT, int WPB, bool H2 = false>
__global__ __launch_bounds__(64 * WPB) void k_lin2(const Lin2Args a) {
    constexpr int L = 32 * NT, CH = L * L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_raw = blockIdx.x * WPB + wave;
    const bool active = tile_raw < a.ntiles;
    const int tile = active ? tile_raw : a.ntiles - 1;
    OPAQUE_LANE();
    RowRef rw = row_of(tile, c, a.rows);
    rw.valid = rw.valid && active;
    CP_PRIME(HW(a.W0));
    f32x16 x[NT], acc[NT];
    load_frag<NT>(x, row_ptr(a.X0, rw.rr, L, h), STRIDE_ROW);
    CP_PREFETCH(HW(a.W1));
    if constexpr (H2) {
        // (fp16 pieces: nothing but the input row lives across a chain -- with x, the accumulator and the partial sum all live the
        // allocator spills 150 registers under two waves per SIMD)
        if (a.X1) {      // merge: OUT0 = ADD + X0 W0 + X1 W1
            h2_chunk_set(acc, x, CP_W(), lane, HRS(a.W0));
            CP_ADVANCE();
            if (a.ADD) add_frag<NT>(acc, row_ptr(a.ADD, rw.rr, L, h), STRIDE_ROW);
            load_frag<NT>(x, row_ptr(a.X1, rw.rr, L, h), STRIDE_ROW);
            h2_chunk_inplace(acc, x, CP_W(), lane, HRS(a.W1));
            if (rw.valid) store_frag<NT>(row_ptr(a.OUT0, rw.row, L, h), STRIDE_ROW, acc);
        } else {         // split: OUT0 = X0 W0, OUT1 = X0 W1
            h2_chunk_set(acc, x, CP_W(), lane, HRS(a.W0));
            CP_ADVANCE();
            if (rw.valid) store_frag<NT>(row_ptr(a.OUT0, rw.row, L, h), STRIDE_ROW, acc);
            h2_chunk_set(acc, x, CP_W(), lane, HRS(a.W1));
            if (rw.valid) store_frag<NT>(row_ptr(a.OUT1, rw.row, L, h), STRIDE_ROW, acc);
        }
        return;
    }
    if (a.X1) {      // merge
        if (a.ADD) load_frag<NT>(acc, row_ptr(a.ADD, rw.rr, L, h), STRIDE_ROW);
        else zero_frag<NT>(acc);
        mfma_chunk<NT, true>(acc, x, CP_W(), lane);
        CP_ADVANCE();
        load_frag<NT>(x, row_ptr(a.X1, rw.rr, L, h), STRIDE_ROW);
        mfma_chunk<NT, true>(acc, x, CP_W(), lane);
        if (rw.valid) store_frag<NT>(row_ptr(a.OUT0, rw.row, L, h), STRIDE_ROW, acc);
    } else {         // split
        zero_frag<NT>(acc);
        mfma_chunk<NT, true>(acc, x, CP_W(), lane);
        CP_ADVANCE();
        if (rw.valid) store_frag<NT>(row_ptr(a.OUT0, rw.row, L, h), STRIDE_ROW, acc);
        zero_frag<NT>(acc);
        mfma_chunk<NT, true>(acc, x, CP_W(), lane);
        if (rw.valid) store_frag<NT>(row_ptr(a.OUT1, rw.row, L, h), STRIDE_ROW, acc);
    }
}

// ================================================================================================
// weight gradient: dW[in][out] = sum_rows X[row][in] * G[row][out] on v_mfma_f32_32x32x2_f32 with the ROW index as the
// reduction dimension: A operand = X^T (lane (m = l&31, k = l>>5) reads X[row 2q+k][32 ti + m], 128 contiguous bytes per
// half wave), B operand = G (lane (n, k) reads G[row 2q+k][32 tj + n]).  Wave ti of a block owns input-feature block ti
// and all NT output blocks; a block owns a contiguous row range and writes its partial dW (and the column sums of G).
// ================================================================================================
#ifndef MGN_WG_ROWS
#define MGN_WG_ROWS 128
#endif
constexpr int WG_ROWS = MGN_WG_ROWS;    // rows per block at least (k_wgrad_lds, cylinder mesh: 64 / 128 / 256 / 512 rows 2.49 / 2.41 / 2.56 / 3.39 ms per step: a block writes a 64 KiB partial whatever its rows)
constexpr int WG_UNROLL = 8;   // k-steps (2 rows each) whose loads are issued together

template <int NT>
__global__ __launch_bounds__(64 * NT) void k_wgrad(const WgradBatch wb) {
    constexpr int L = 32 * NT;
    const WgradJob& jb = wb.job[blockIdx.y];
    const int64_t r0 = (int64_t)blockIdx.x * wb.rows_per_block;
    if (r0 >= jb.rows) return;
    const int64_t r1 = r0 + wb.rows_per_block < jb.rows ? r0 + wb.rows_per_block : jb.rows;
    const int lane = threadIdx.x & 63, m = lane & 31, kk = lane >> 5;
    const int ti = threadIdx.x >> 6;
    const float* __restrict__ X = jb.X;
    const float* __restrict__ G = jb.G;
    const int32_t* __restrict__ xidx = jb.xidx;
    const bool with_w = jb.pw != nullptr;
    if (!with_w && ti != 0) return;              // column sums only: one wave
    f32x16 acc[NT];
    float bs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bs[t] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    }
    for (int64_t q = r0; q < r1; q += 2 * WG_UNROLL) {
        float av[WG_UNROLL], bv[WG_UNROLL][NT];
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
            const int64_t row = q + 2 * u + kk;
            const bool ok = row < r1;
            const int64_t rr = ok ? row : r0;
            float xa = 0.f;
            if (with_w) {
                const int64_t src = xidx ? (int64_t)xidx[rr] : rr;
                xa = X[src * L + 32 * ti + m];
            }
            av[u] = ok ? xa : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float gb = G[rr * L + 32 * t + m];
                bv[u][t] = ok ? gb : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (with_w) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][t], acc[t], 0, 0, 0);
                bs[t] += bv[u][t];
            }
    }
    // D layout of the 32x32 MFMA: register r of lane l holds D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]
    if (with_w) {
        float* pw = jb.pw + (size_t)blockIdx.x * L * L;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) pw[(size_t)(32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kk) * L + 32 * t + m] = acc[t][r];
    }
    if (ti == 0 && jb.pb) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float sb = bs[t] + __shfl_xor(bs[t], 32, 64);
            if (kk == 0) jb.pb[(size_t)blockIdx.x * L + 32 * t + m] = sb;
        }
    }
}

// The same product at L = 128 with the rows staged through LDS (round 5, second session).  k_wgrad's operands are 4-byte loads -- 40
// vector-memory instructions of 256 bytes per wave and 16 rows, none of them in flight while the 32 MFMAs of those rows run: on the
// cylinder mesh a block's row loop is a chain of L2 round trips (27 us per launch for 8 us of MFMAs), on M-1M the launch is bound by the
// issue of those instructions.  Here the block's four waves fetch a chunk of 16 rows of X and of G with 16-byte loads (4 instructions per
// wave), put it into LDS (row stride 132 floats: a column read meets 32 banks) and read the MFMA operands from there; the next chunk's
// two chunks' global loads are in flight during a chunk's MFMAs (32 registers), row indices two chunks ahead of their rows.  Same MFMAs in the
// same order as k_wgrad: the same bits.
constexpr int WGL_ROWS = 16, WGL_LS = 132;
__global__ __launch_bounds__(256) void k_wgrad_lds(const WgradBatch wb) {
    constexpr int NT = 4, L = 128;
    __shared__ __attribute__((aligned(16))) float sX[2][WGL_ROWS * WGL_LS];
    __shared__ __attribute__((aligned(16))) float sG[2][WGL_ROWS * WGL_LS];
    const WgradJob& jb = wb.job[blockIdx.y];
    const int64_t r0 = (int64_t)blockIdx.x * wb.rows_per_block;
    if (r0 >= jb.rows) return;
    const int64_t r1 = r0 + wb.rows_per_block < jb.rows ? r0 + wb.rows_per_block : jb.rows;
    const int lane = threadIdx.x & 63, m = lane & 31, kk = lane >> 5;
    const int ti = threadIdx.x >> 6;
    const float* __restrict__ X = jb.X;
    const float* __restrict__ G = jb.G;
    const int32_t* __restrict__ xidx = jb.xidx;
    const bool with_w = jb.pw != nullptr;
    f32x16 acc[NT];
    float bs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bs[t] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    }
    // loader role: thread (lr = tid / 32, lc = tid % 32) moves 16 bytes of rows lr and lr + 8 of a chunk.  TWO chunks of global loads are in
    // flight (one chunk of MFMAs is ~1 us: less than an HBM round trip under load), the X row indices of a chunk are read two chunks before
    // its rows.
    const int lr = threadIdx.x >> 5, lc = threadIdx.x & 31;
    const int nchunks = (int)((r1 - r0 + WGL_ROWS - 1) / WGL_ROWS);
    auto rows_of = [&](int c, int64_t (&out)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int64_t row = r0 + (int64_t)c * WGL_ROWS + lr + 8 * p;
            const int64_t rr = (c < nchunks && row < r1) ? row : r0;
            out[p] = (with_w && xidx) ? (int64_t)xidx[rr] : rr;
        }
    };
    auto fetch = [&](int c, const int64_t (&src)[2], f32x4 (&xr)[2], f32x4 (&gr)[2]) {   // chunk c -> registers (rows past the end: row r0, zeroed when stored)
        if (c >= nchunks) return;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int64_t row = r0 + (int64_t)c * WGL_ROWS + lr + 8 * p;
            const int64_t rr = row < r1 ? row : r0;
            if (with_w) xr[p] = reinterpret_cast<const f32x4*>(X + src[p] * L)[lc];
            gr[p] = reinterpret_cast<const f32x4*>(G + rr * L)[lc];
        }
    };
    auto to_lds = [&](int c, int b, const f32x4 (&xr)[2], const f32x4 (&gr)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool ok = r0 + (int64_t)c * WGL_ROWS + lr + 8 * p < r1;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (with_w) *reinterpret_cast<f32x4*>(&sX[b][(lr + 8 * p) * WGL_LS + 4 * lc]) = ok ? xr[p] : z;
            *reinterpret_cast<f32x4*>(&sG[b][(lr + 8 * p) * WGL_LS + 4 * lc]) = ok ? gr[p] : z;
        }
    };
    auto compute = [&](int b) {
        if (!(with_w || ti == 0)) return;
#pragma unroll
        for (int u = 0; u < WGL_ROWS / 2; ++u) {
            const int row = 2 * u + kk;
            const float av = with_w ? sX[b][row * WGL_LS + 32 * ti + m] : 0.f;
            float bv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = sG[b][row * WGL_LS + 32 * t + m];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (with_w) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
                bs[t] += bv[t];
            }
        }
    };
    int64_t sa[2], sb2[2];                               // X rows of the chunks in slots A / B (then: of the chunks that follow them there)
    f32x4 xa[2], ga[2], xb[2], gb[2];
    rows_of(0, sa);
    rows_of(1, sb2);
    fetch(0, sa, xa, ga);
    fetch(1, sb2, xb, gb);
    rows_of(2, sa);
    rows_of(3, sb2);
    for (int c = 0; c < nchunks; c += 2) {
        to_lds(c, 0, xa, ga);
        __syncthreads();          // chunk c is in LDS; every wave is past its reads of chunk c - 1 (the buffer chunk c + 1 goes to)
        fetch(c + 2, sa, xa, ga);
        rows_of(c + 4, sa);
        compute(0);
        if (c + 1 >= nchunks) break;
        to_lds(c + 1, 1, xb, gb);
        __syncthreads();
        fetch(c + 3, sb2, xb, gb);
        rows_of(c + 5, sb2);
        compute(1);
    }
    // D layout of the 32x32 MFMA: register r of lane l holds D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]
    if (with_w) {
        float* pw = jb.pw + (size_t)blockIdx.x * L * L;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) pw[(size_t)(32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kk) * L + 32 * t + m] = acc[t][r];
    }
    if (ti == 0 && jb.pb) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float sb = bs[t] + __shfl_xor(bs[t], 32, 64);
            if (kk == 0) jb.pb[(size_t)blockIdx.x * L + 32 * t + m] = sb;
        }
    }
}

// Round 6: the same product on v_mfma_f32_32x32x16_f16 -- X and G as two fp16 pieces each (split_common.hpp), three piece products, fp32
// accumulation.  The fp32 instruction above runs a 16-row k-block of one 32 x 32 output block in 8 x 64 cycles, the 16-bit one in 3 x 32: the
// launch is left with its bytes (M-1M) and its latencies (cylinder mesh).  Rows are the reduction dimension, so one scale per operand has to
// hold for all rows of an MFMA: a chunk of 32 rows (two k-steps) is scaled by the power of two of its own largest magnitude (per array: all
// 128 features); the block's sum is kept in the units of the largest scales met so far (`rescale`) and un-scaled once, at the end.  The MFMA wants
// eight ROWS of one feature per lane: 128 threads per array fetch (8 rows) x (4 features) each with eight 16-byte loads, split their 32
// values and write four 16-byte units (feature f, row block kb) per piece -- the transposition happens in the registers of the loader.
// LDS unit order within a row block: (f & 3) * 36 + (f >> 2): both the loaders' writes (features 4 lc + i, lc = 0..31) and the operand
// reads (features 32 t + m, m = 0..31) meet every bank once.  Two barriers per chunk (its scale, its pieces), one LDS buffer; a chunk's
// rows are requested two chunks ahead.  Same partial-block format as k_wgrad (k_reduce_partials sums them in order).
constexpr int WH_ROWS = 32;                 // rows per chunk
constexpr int WH_KB = WH_ROWS / 8;          // row blocks (one 16-byte unit of eight fp16 per feature and row block)
constexpr int WH_UPK = 4 * 36;              // units per row block
constexpr int WH_BUF = WH_KB * WH_UPK;      // units per (array, piece) of a buffer: 9 216 bytes
constexpr size_t WH_LDS = (size_t)4 * WH_BUF * 16 + 4 * sizeof(float) + 4 * 128 * sizeof(float);
DEVINL int wh_unit(int f, int kb) { return kb * WH_UPK + (f & 3) * 36 + (f >> 2); }

#ifndef MGN_WH_WHATIF
#define MGN_WH_WHATIF 0        // timing builds (wrong gradients): 1 no products, 2 no split / LDS writes, 4 no partial store, 8 no maxima, 16 no loads after the first two chunks
#endif
#ifndef MGN_WH_BLOCKS
#define MGN_WH_BLOCKS 2         // blocks per CU (3: the operand reads of a chunk no longer fit the 168 registers -- 317 spilled)
#endif
__global__ __launch_bounds__(256, MGN_WH_BLOCKS) void k_wgrad_h2(const WgradBatch wb) {
    constexpr int NT = 4, L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* const sP = reinterpret_cast<u32x4*>(smem);                       // [X hi, X lo, G hi, G lo][WH_BUF]
    float* const smax = smem + (size_t)4 * WH_BUF * 4;                      // [wave]: waves 0, 1 load X, waves 2, 3 load G
    float* const sbs = smax + 4;                                             // [row group][feature]: column sums of G
    const WgradJob& jb = wb.job[blockIdx.y];
    const int64_t r0 = (int64_t)blockIdx.x * wb.rows_per_block;
    if (r0 >= jb.rows) return;
    const int64_t r1 = r0 + wb.rows_per_block < jb.rows ? r0 + wb.rows_per_block : jb.rows;
    const int lane = threadIdx.x & 63, m = lane & 31, kg = lane >> 5;
    const int ti = threadIdx.x >> 6;
    const float* __restrict__ X = jb.X;
    const float* __restrict__ G = jb.G;
    const int32_t* __restrict__ xidx = jb.xidx;
    const bool with_w = jb.pw != nullptr;
    if (jb.Y) {
        // LayerNorm job: dbeta = column sums of g, dgamma = column sums of g * xhat, with g and xhat rebuilt from the arrays the backward
        // kernel read (G0 (+ G1), Y) and the row statistics it left -- the same operations in the same order as there.  Thread (rl, lc):
        // rows r0 + rl, r0 + rl + 8, ..., features 4 lc .. 4 lc + 3, four rows in flight; the eight row lanes are added in a fixed order.
        const int rl = threadIdx.x >> 5, lc4 = threadIdx.x & 31;
        const float* __restrict__ Y = jb.Y;
        const float* __restrict__ G1 = jb.G1;
        const int32_t* __restrict__ g1i = jb.g1idx;
        const float2* __restrict__ ST = reinterpret_cast<const float2*>(jb.LNROW);
        f32x4 sb = {0.f, 0.f, 0.f, 0.f}, sg = {0.f, 0.f, 0.f, 0.f};
        for (int64_t row = r0 + rl; row < r1; row += 32) {
            f32x4 gv[4], yv[4], g1v[4];
            float2 st[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = row + 8 * u < r1 ? row + 8 * u : r0;
                gv[u] = reinterpret_cast<const f32x4*>(G + rr * L)[lc4];
                yv[u] = reinterpret_cast<const f32x4*>(Y + rr * L)[lc4];
                st[u] = ST[rr];
                if (G1) g1v[u] = reinterpret_cast<const f32x4*>(G1 + (g1i ? (int64_t)g1i[rr] : rr) * L)[lc4];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (row + 8 * u >= r1) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gg = G1 ? gv[u][i] + g1v[u][i] : gv[u][i];
                    // (one value at a time, fenced: as packed fp32 instructions with op_sel / neg on the loaded (mean, 1 / denominator) pair --
                    // what hipcc makes of the plain expression, and of an fma form -- the sums of features 4 lc, 4 lc + 2, lc >= 16 differed from
                    // run to run by ~1e-5 of their value, on every MLP alike; Y alone or the pair alone: stable.  docs/experiments.md)
                    float d = yv[u][i] - st[u].x;
                    asm volatile("" : "+v"(d));
                    float xh = d * st[u].y;
                    asm volatile("" : "+v"(xh));
                    sb[i] += gg;
                    sg[i] += gg * xh;
                }
            }
        }
        float* const red = smem;                          // [2][8 row lanes][L]
        *reinterpret_cast<f32x4*>(&red[rl * L + 4 * lc4]) = sb;
        *reinterpret_cast<f32x4*>(&red[(8 + rl) * L + 4 * lc4]) = sg;
        __syncthreads();
        {
            const int which = threadIdx.x >> 7, f = threadIdx.x & 127;
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += red[(8 * which + q) * L + f];
            float* const out = which ? jb.pb2 : jb.pb;
            if (out) out[(size_t)blockIdx.x * L + f] = sum;
        }
        return;
    }
    // loader role: array (X: threads 0..127, G: 128..255), row group rg (8 rows of the chunk), lc: features 4 lc .. 4 lc + 3
    const int arr = threadIdx.x >> 7, rg = (threadIdx.x >> 5) & 3, lc = threadIdx.x & 31;
    const bool loads = arr == 1 || with_w;
    const int nchunks = (int)((r1 - r0 + WH_ROWS - 1) / WH_ROWS);
    f32x16 acc[NT];
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

    auto fetch = [&](int c, f32x4 (&d)[8]) {             // chunk c -> registers (rows past the end: row r0, zeroed in `amax_zero`)
        if (c >= nchunks || !loads) return;
        if (MGN_WH_WHATIF & 16) { if (c > 1) return; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t row = r0 + (int64_t)c * WH_ROWS + 8 * rg + j;
            const int64_t rr = row < r1 ? row : r0;
            if (arr == 0) {
                const int64_t src = xidx ? (int64_t)xidx[rr] : rr;
                d[j] = reinterpret_cast<const f32x4*>(X + src * L)[lc];
            } else {
                d[j] = reinterpret_cast<const f32x4*>(G + rr * L)[lc];
            }
        }
    };
    // zero the rows past the end, then: this wave's largest magnitude of chunk c -> its slot
    auto amax_zero = [&](int c, f32x4 (&d)[8]) {
        if (c >= nchunks) return;
        if (MGN_WH_WHATIF & 8) { if (lane == 0 && c == 0) smax[ti] = 1.f; return; }
        float mx = 0.f;
        if (loads) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = r0 + (int64_t)c * WH_ROWS + 8 * rg + j < r1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[j][i] = ok ? d[j][i] : 0.f;
                    mx = __builtin_fmaxf(mx, __builtin_fabsf(d[j][i]));
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) smax[ti] = mx;
    };
    auto to_lds = [&](const f32x4 (&d)[8], float s) {
        if (!loads || (MGN_WH_WHATIF & 2)) return;
        u32x4* const ph = sP + (size_t)(2 * arr) * WH_BUF;
        u32x4* const pl = ph + WH_BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) h2_split_pair<0>(hi[j], lo[j], d[2 * j][i], d[2 * j + 1][i], s);
            const int u = wh_unit(4 * lc + i, rg);
            ph[u] = u32x4{hi[0], hi[1], hi[2], hi[3]};
            pl[u] = u32x4{lo[0], lo[1], lo[2], lo[3]};
            if (arr == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bs[i] += d[j][i];
            }
        }
    };
    auto compute = [&]() {
        if (!with_w || (MGN_WH_WHATIF & 1)) return;
        const u32x4* const xh = sP;
#pragma unroll
        for (int ks = 0; ks < WH_ROWS / 16; ++ks) {
            const int ua = wh_unit(32 * ti + m, 2 * ks + kg);
            const sp_f16x8 ah = h2_wop(xh[ua]), al = h2_wop(xh[WH_BUF + ua]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int ub = wh_unit(32 * t + m, 2 * ks + kg);
                const sp_f16x8 bh = h2_wop(xh[2 * WH_BUF + ub]), bl = h2_wop(xh[3 * WH_BUF + ub]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);      // small terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    };
    // the running scales: the power of two of the largest magnitude met so far in each array.  When a chunk raises one, the sum so far
    // moves to the new units (a power of two <= 1: exact up to underflow of terms 2^-100 below the new largest ones).
    float runx = 0.f, rung = 0.f;                         // largest magnitudes so far
    H2Scale sx = h2_scale(0.f), sg = h2_scale(0.f);
    auto rescale = [&]() {
        const float mx = __builtin_fmaxf(runx, __builtin_fmaxf(smax[0], smax[1])), mg = __builtin_fmaxf(rung, __builtin_fmaxf(smax[2], smax[3]));
        runx = mx;
        rung = mg;
        const H2Scale nx = h2_scale(mx), ng = h2_scale(mg);
        const float ratio = (nx.s * sx.rs) * (ng.s * sg.rs);      // new units / old units
        sx = nx;
        sg = ng;
        if (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ratio)) != 0x3f800000) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] *= ratio;
        }
    };
    f32x4 da[8], db[8];                                   // chunks c and c + 1 (then: the chunks that follow them in their slots)
    fetch(0, da);
    fetch(1, db);
    // two barriers per chunk (scale, pieces): a chunk's rows are requested two chunks before their first use -- with the scale published a
    // chunk ahead (one barrier) they were used one chunk after their request and every chunk waited for the memory round trip
    for (int c = 0; c < nchunks; c += 2) {
        amax_zero(c, da);
        __syncthreads();          // the scale of chunk c; every wave is past its reads of chunk c - 1's pieces
        rescale();
        to_lds(da, arr ? sg.s : sx.s);
        fetch(c + 2, da);
        __syncthreads();          // (every wave has read the maxima: the next chunk's may be written)
        compute();
        if (c + 1 >= nchunks) break;
        amax_zero(c + 1, db);
        __syncthreads();
        rescale();
        to_lds(db, arr ? sg.s : sx.s);
        fetch(c + 3, db);
        __syncthreads();
        compute();
    }
    // D layout of the 32x32 MFMA: register r of lane l holds D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]
    if (with_w && !(MGN_WH_WHATIF & 4)) {
        const float cc = sx.rs * sg.rs;                   // back from the scaled units
        float* pw = jb.pw + (size_t)blockIdx.x * L * L;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) pw[(size_t)(32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kg) * L + 32 * t + m] = acc[t][r] * cc;
    }
    if (jb.pb) {                                          // column sums of G: the four row groups' sums in a fixed order
        if (arr == 1) *reinterpret_cast<f32x4*>(&sbs[rg * L + 4 * lc]) = f32x4{bs[0], bs[1], bs[2], bs[3]};
        __syncthreads();
        if (threadIdx.x < L) {
            const int f = threadIdx.x;
            jb.pb[(size_t)blockIdx.x * L + f] = ((sbs[f] + sbs[L + f]) + sbs[2 * L + f]) + sbs[3 * L + f];
        }
    }
}

#ifndef MGN_REDUCE_VEC
#define MGN_REDUCE_VEC 0     // 1: four outputs per thread, eight blocks in flight -- cylinder-mesh step 2.44 ms against 2.38 (a quarter of the threads: fewer loads in flight overall)
#endif
// out[r * cols + c] = sum_b partial[b][r * ld + c], fixed order (bitwise reproducible); one job per blockIdx.y.  Four consecutive outputs per
// thread where the shapes allow 16-byte accesses (cols = ld, a multiple of 4: every weight chunk and bias of the model but the decoder's
// last layer), eight partial blocks in flight (-DMGN_REDUCE_VEC=1; measured slower, off).
__global__ void k_reduce_partials(const ReduceBatch rb) {
    const ReduceJob& jb = rb.job[blockIdx.y];
    const int total = jb.nrows * jb.cols;
    if (MGN_REDUCE_VEC && jb.cols == jb.ld && (jb.cols & 3) == 0 && (jb.block_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(jb.out) | reinterpret_cast<uintptr_t>(jb.partial)) & 15) == 0) {
        const int i4 = blockIdx.x * blockDim.x + threadIdx.x;
        if (4 * i4 >= total) return;
        const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(jb.partial) + i4;
        const size_t bs4 = (size_t)jb.block_stride / 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int b = 0;
        for (; b + 8 <= jb.nblocks; b += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(b + u) * bs4];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < jb.nblocks; ++b) s += p[(size_t)b * bs4];
        reinterpret_cast<f32x4*>(jb.out)[i4] = s;
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = i / jb.cols, cidx = i - r * jb.cols;
    const float* __restrict__ p = jb.partial + (size_t)r * jb.ld + cidx;
    float s = 0.f;
    int b = 0;
    for (; b + 4 <= jb.nblocks; b += 4) {        // independent loads, same summation order
        const float v0 = p[(size_t)b * jb.block_stride], v1 = p[(size_t)(b + 1) * jb.block_stride];
        const float v2 = p[(size_t)(b + 2) * jb.block_stride], v3 = p[(size_t)(b + 3) * jb.block_stride];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; b < jb.nblocks; ++b) s += p[(size_t)b * jb.block_stride];
    jb.out[i] = s;
}

__global__ void k_segment_sum(const float* __restrict__ src, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
                              const float* __restrict__ add, float* __restrict__ out, int32_t n, int L4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * L4) return;
    const int node = (int)(i / L4), q = (int)(i - (int64_t)node * L4);
    f32x4 s = add ? reinterpret_cast<const f32x4*>(add)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* S4 = reinterpret_cast<const f32x4*>(src);
    int p = rowptr[node];
    const int e1 = rowptr[node + 1];
    for (; p + 4 <= e1; p += 4) {                        // (four rows in flight, added in order: see k_segment_sum2)
        const int64_t r0 = perm ? perm[p] : p, r1 = perm ? perm[p + 1] : p + 1, r2 = perm ? perm[p + 2] : p + 2, r3 = perm ? perm[p + 3] : p + 3;
        const f32x4 v0 = S4[r0 * L4 + q], v1 = S4[r1 * L4 + q], v2 = S4[r2 * L4 + q], v3 = S4[r3 * L4 + q];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; p < e1; ++p) {
        const int64_t row = perm ? perm[p] : p;
        s += S4[row * L4 + q];
    }
    reinterpret_cast<f32x4*>(out)[i] = s;
}

// out[n] = add[n] + sum over the receiver range of srcA + sum over the (permuted) sender range of srcB, in that order: the
// two segmented sums of a processor step's backward in one launch (same summation order as two k_segment_sum passes)
__global__ void k_segment_sum2(const float* __restrict__ srcA, const int32_t* __restrict__ rowptrA, const float* __restrict__ srcB,
                               const int32_t* __restrict__ rowptrB, const int32_t* __restrict__ permB, const float* add, float* out, int32_t n,
                               int L4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * L4) return;
    const int node = (int)(i / L4), q = (int)(i - (int64_t)node * L4);
    f32x4 s = reinterpret_cast<const f32x4*>(add)[i];
    // four rows requested together, added in their order (a mesh node has ~6 rows per list: as a plain loop the launch was a chain of
    // dependent L2 round trips, 19.6 us on the cylinder mesh)
    const f32x4* A4 = reinterpret_cast<const f32x4*>(srcA);
    const f32x4* B4 = reinterpret_cast<const f32x4*>(srcB);
    int p = rowptrA[node];
    const int ea = rowptrA[node + 1];
    for (; p + 4 <= ea; p += 4) {
        const f32x4 v0 = A4[(int64_t)p * L4 + q], v1 = A4[(int64_t)(p + 1) * L4 + q], v2 = A4[(int64_t)(p + 2) * L4 + q], v3 = A4[(int64_t)(p + 3) * L4 + q];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; p < ea; ++p) s += A4[(int64_t)p * L4 + q];
    p = rowptrB[node];
    const int eb = rowptrB[node + 1];
    for (; p + 4 <= eb; p += 4) {
        const int r0 = permB[p], r1 = permB[p + 1], r2 = permB[p + 2], r3 = permB[p + 3];
        const f32x4 v0 = B4[(int64_t)r0 * L4 + q], v1 = B4[(int64_t)r1 * L4 + q], v2 = B4[(int64_t)r2 * L4 + q], v3 = B4[(int64_t)r3 * L4 + q];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; p < eb; ++p) s += B4[(int64_t)permB[p] * L4 + q];
    reinterpret_cast<f32x4*>(out)[i] = s;
}

__global__ void k_segment_sum_pair(const float* __restrict__ src, const int32_t* __restrict__ rowptr_r, const int32_t* __restrict__ rowptr_s,
                                   const int32_t* __restrict__ perm_s, float* __restrict__ out_r, float* __restrict__ out_s, int32_t n, int L4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * L4) return;
    const int node = (int)(i / L4), q = (int)(i - (int64_t)node * L4);
    const f32x4* S4 = reinterpret_cast<const f32x4*>(src);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int p = rowptr_r[node];
    const int er = rowptr_r[node + 1];
    for (; p + 4 <= er; p += 4) {                        // (four rows in flight, added in order: see k_segment_sum2)
        const f32x4 v0 = S4[(int64_t)p * L4 + q], v1 = S4[(int64_t)(p + 1) * L4 + q], v2 = S4[(int64_t)(p + 2) * L4 + q], v3 = S4[(int64_t)(p + 3) * L4 + q];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; p < er; ++p) s += S4[(int64_t)p * L4 + q];
    reinterpret_cast<f32x4*>(out_r)[i] = s;
    s = f32x4{0.f, 0.f, 0.f, 0.f};
    p = rowptr_s[node];
    const int es = rowptr_s[node + 1];
    for (; p + 4 <= es; p += 4) {
        const int r0 = perm_s[p], r1 = perm_s[p + 1], r2 = perm_s[p + 2], r3 = perm_s[p + 3];
        const f32x4 v0 = S4[(int64_t)r0 * L4 + q], v1 = S4[(int64_t)r1 * L4 + q], v2 = S4[(int64_t)r2 * L4 + q], v3 = S4[(int64_t)r3 * L4 + q];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; p < es; ++p) s += S4[(int64_t)perm_s[p] * L4 + q];
    reinterpret_cast<f32x4*>(out_s)[i] = s;
}

// dst [rows][L] = [ (srcA | srcB)[gid ? gid[row] : row][wa + wb] * scale + shift | 0 ]   (scale == null: identity)
__global__ void k_affine_pad(const float* __restrict__ srcA, int wa, const float* __restrict__ srcB, int wb, const float* __restrict__ scale,
                             const float* __restrict__ shift, const int32_t* __restrict__ gid, float* __restrict__ dst, int L, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * L) return;
    const int64_t r = i / L;
    const int f = (int)(i - r * L);
    float v = 0.f;
    if (f < wa + wb) {
        const int64_t sr = gid ? (int64_t)gid[r] : r;          // gid: dst row r <- source row gid[r] (the caller's order -> the engine's)
        v = f < wa ? srcA[sr * wa + f] : srcB[sr * wb + (f - wa)];
        if (scale) v = v * scale[f] + shift[f];
    }
    dst[i] = v;
}

// rows of a [rows][width] array between the caller's node order and the engine's (gid[i] = caller's row of engine row i):
// gather: dst[i] = src[gid[i]];  scatter: dst[gid[i]] = src[i]
__global__ void k_permute_rows(float* __restrict__ dst, const float* __restrict__ src, const int32_t* __restrict__ gid, int64_t rows, int width,
                               int scatter) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * width) return;
    const int64_t r = i / width;
    const int f = (int)(i - r * width);
    const int64_t g = gid[r];
    if (scatter) dst[g * width + f] = src[i];
    else dst[i] = src[g * width + f];
}

// ---- whole-array LayerNorm (mgn_config.ln_dims = MGN_LN_ALL): statistics over ALL rows x L values of an MLP's output, then apply ----
// partial[b] = (sum, sum of squares) of block b's grid-strided share, in double, lanes and waves added in a fixed order
__global__ void k_array_stats(const float* __restrict__ x, int64_t n, double* __restrict__ partial) {
    __shared__ double sh[2][4];
    double s = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)x[i];
        s += v;
        q += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); q += __shfl_down(q, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        partial[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}
// stats = (mean, 1 / (sqrt(var + eps_in) + eps_out), kappa = (sqrt(var + eps_in) + eps_out) / sqrt(var + eps_in) -- the factor of the xhat term
// of the pullback): biased variance over the n values, blocks added in order
__global__ __launch_bounds__(1024) void k_array_stats_final(const double* __restrict__ partial, int64_t nb, int64_t n, float eps_in, float eps_out,
                                                            float* __restrict__ stats) {
    // thread t adds slots t, t + 1024, ... in order; a fixed tree adds the 1 024 sums: the same bits whatever ran first (as ONE thread
    // over 1 024 partials this launch took ~0.1 ms -- more than the MLP kernel it follows on the cylinder mesh)
    __shared__ double sh[2][1024];
    double s = 0.0, q = 0.0;
    for (int64_t b = threadIdx.x; b < nb; b += 1024) { s += partial[2 * b]; q += partial[2 * b + 1]; }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {                // a fixed tree over the 1 024 sums
        if ((int)threadIdx.x < w) { sh[0][threadIdx.x] += sh[0][threadIdx.x + w]; sh[1][threadIdx.x] += sh[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;                       // (no barrier below)
    s = sh[0][0]; q = sh[1][0];
    const double mean = n > 0 ? s / (double)n : 0.0;
    double var = n > 0 ? q / (double)n - mean * mean : 0.0;
    if (var < 0.0) var = 0.0;
    const double lsq = sqrt(var + (double)eps_in);
    stats[0] = (float)mean;
    stats[1] = (float)(1.0 / (lsq + (double)eps_out));
    stats[2] = lsq > 0.0 ? (float)((lsq + (double)eps_out) / lsq) : 1.f;
}
// t = (y - mean) * rden * gamma[f] + beta[f];  lnout = t;  out = (resid ? resid : 0) + t   (out may alias resid, lnout may alias y)
__global__ void k_ln_all_apply(const float* y, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* resid, float* out, float* lnout, int64_t n, int L) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = (int)(i % L);
    const float t = (y[i] - stats[0]) * stats[1] * gamma[f] + beta[f];
    const float r = resid ? resid[i] : 0.f;
    if (lnout) lnout[i] = t;
    if (out) out[i] = r + t;
}

// the edge half of a processor step under the whole-array LayerNorm in one pass over the receiver CSR: t = LN(y[p]) (statistics given),
// e[p] += t, agg[n] = sum of t over the edges n receives, added in edge order (k_segment_sum's order)
__global__ void k_ln_all_apply_segsum(const float* __restrict__ y, const float* __restrict__ stats, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float* __restrict__ e, const int32_t* __restrict__ rowptr,
                                      float* __restrict__ agg, int32_t n, int L4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * L4) return;
    const int node = (int)(i / L4), q = (int)(i - (int64_t)node * L4);
    const float mean = stats[0], rden = stats[1];
    const f32x4 g4 = reinterpret_cast<const f32x4*>(gamma)[q], b4 = reinterpret_cast<const f32x4*>(beta)[q];
    const f32x4* Y4 = reinterpret_cast<const f32x4*>(y);
    f32x4* E4 = reinterpret_cast<f32x4*>(e);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    auto one = [&](f32x4 v, f32x4 ev, int64_t p) {
        f32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (v[j] - mean) * rden * g4[j] + b4[j];
        E4[p * L4 + q] = ev + t;
        s += t;
    };
    int p = rowptr[node];
    const int e1 = rowptr[node + 1];
    for (; p + 2 <= e1; p += 2) {                        // two rows of each stream in flight, added in order
        const f32x4 v0 = Y4[(int64_t)p * L4 + q], v1 = Y4[(int64_t)(p + 1) * L4 + q];
        const f32x4 a0 = E4[(int64_t)p * L4 + q], a1 = E4[(int64_t)(p + 1) * L4 + q];
        one(v0, a0, p);
        one(v1, a1, p + 1);
    }
    for (; p < e1; ++p) one(Y4[(int64_t)p * L4 + q], E4[(int64_t)p * L4 + q], p);
    reinterpret_cast<f32x4*>(agg)[i] = s;
}

// ---- reverse pass of the whole-array LayerNorm ----
// t = gamma xhat + beta with xhat = (y - mean) rden over ALL rows x L values.  With G the gradient w.r.t. t (G = G0[row] (+ G1[g1idx[row]])):
//   dbeta[f] = sum_rows G,  dgamma[f] = sum_rows G xhat,
//   dy = rden (gamma G - m1 - xhat m2),  m1 = mean_all(gamma G) = sum_f gamma[f] dbeta[f] / n,  m2 = kappa mean_all(gamma G xhat) = kappa sum_f gamma[f] dgamma[f] / n
// k_lnall_bwd_cols: per block the column sums of its rows in double (block b: rows 2 b + {0, 1}, stride 2 gridDim; fixed order);
// k_lnall_bwd_cols_final: blocks added in order -> dbeta, dgamma (into the gradient vector), m = (m1, m2); the map G -> dy itself runs inside
// the MLP backward kernels as they load G (TrainBwdArgs::ln = 2).
__global__ __launch_bounds__(256) void k_lnall_bwd_cols(const float* __restrict__ G0, const float* __restrict__ G1, const int32_t* __restrict__ g1idx,
                                                        const float* __restrict__ Y, const float* __restrict__ stats, int64_t rows, int L,
                                                        double* __restrict__ partial) {
    __shared__ double sh[2][128];
    const int f = threadIdx.x & 127, half = threadIdx.x >> 7;
    const float mean = stats[0], rden = stats[1];
    double sb = 0.0, sg = 0.0;
    if (f < L)
        for (int64_t r = (int64_t)blockIdx.x * 2 + half; r < rows; r += (int64_t)gridDim.x * 2) {
            float g = G0[r * L + f];
            if (G1) g += G1[(g1idx ? (int64_t)g1idx[r] : r) * L + f];
            const float xh = (Y[r * L + f] - mean) * rden;
            sb += (double)g;
            sg += (double)(g * xh);
        }
    if (half == 1) { sh[0][f] = sb; sh[1][f] = sg; }
    __syncthreads();
    if (half == 0) {
        partial[((size_t)blockIdx.x * 2 + 0) * 128 + f] = sb + sh[0][f];
        partial[((size_t)blockIdx.x * 2 + 1) * 128 + f] = sg + sh[1][f];
    }
}
__global__ __launch_bounds__(1024) void k_lnall_bwd_cols_final(const double* __restrict__ partial, int nb, int64_t n, int L, const float* __restrict__ gamma,
                                                               const float* __restrict__ stats, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ m) {
    // group g (of 8) adds blocks g, g + 8, ... in order per feature; then feature f adds its 8 group sums in order; then thread 0 the features
    __shared__ double sh[2][8][128];
    const int f = threadIdx.x & 127, g = threadIdx.x >> 7;
    double sb = 0.0, sg = 0.0;
    for (int b = g; b < nb; b += 8) { sb += partial[((size_t)b * 2 + 0) * 128 + f]; sg += partial[((size_t)b * 2 + 1) * 128 + f]; }
    sh[0][g][f] = sb;
    sh[1][g][f] = sg;
    __syncthreads();
    if (g == 0) {
        sb = 0.0; sg = 0.0;
        for (int k = 0; k < 8; ++k) { sb += sh[0][k][f]; sg += sh[1][k][f]; }
        if (f < L) { dbeta[f] = (float)sb; dgamma[f] = (float)sg; }
    }
    __syncthreads();
    if (g == 0) {
        sh[0][0][f] = f < L ? sb * (double)gamma[f] : 0.0;
        sh[1][0][f] = f < L ? sg * (double)gamma[f] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, c = 0.0;
        for (int i = 0; i < 128; ++i) { a += sh[0][0][i]; c += sh[1][0][i]; }
        m[0] = n > 0 ? (float)(a / (double)n) : 0.f;
        m[1] = n > 0 ? (float)((double)stats[2] * c / (double)n) : 0.f;
    }
}

// epilogue of a right-hand side (reference src/solve.jl:203-218): out[i][o] = (Y[i][o] os[o] + osh[o]) * (mask ? mask[gid ? gid[i] : i] : 1)
__global__ void k_rhs_epilogue(const float* __restrict__ Y, int L, int O, const float* __restrict__ os, const float* __restrict__ osh,
                               const float* __restrict__ mask, const int32_t* __restrict__ gid, float* __restrict__ out, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * O) return;
    const int64_t n = i / O;
    const int o = (int)(i - n * O);
    float y = Y[n * L + o];
    if (os) y = y * os[o] + osh[o];
    if (mask) y *= mask[gid ? (int64_t)gid[n] : n];
    out[i] = y;
}

// seed of the RHS VJP: dx/dt = (out * os + osh) .* val_mask  =>  G[n][o] = lambda[n][o] * val_mask[n] * os[o]
__global__ void k_vjp_seed(const float* __restrict__ Y, int L, int O, const float* __restrict__ lambda, const float* __restrict__ vm,
                           const float* __restrict__ os, const float* __restrict__ osh, float* __restrict__ G, float* __restrict__ dxdt,
                           int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * O) return;
    const int64_t n = i / O;
    const int o = (int)(i - n * O);
    const float m = vm ? vm[n] : 1.f, sc = os ? os[o] : 1.f, sh = osh ? osh[o] : 0.f;
    G[n * L + o] = lambda[i] * m * sc;
    if (dxdt) dxdt[i] = (Y[n * L + o] * sc + sh) * m;
}

__global__ void k_extract_cols(const float* __restrict__ src, int L, int O, const float* __restrict__ scale, float* __restrict__ dst, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * O) return;
    const int64_t n = i / O;
    const int o = (int)(i - n * O);
    dst[i] = src[n * L + o] * (scale ? scale[o] : 1.f);
}

__global__ void k_loss(const float* __restrict__ Y, int L, const float* __restrict__ target, int O, const int32_t* __restrict__ mask,
                       int64_t nmask, int32_t index_base, float* __restrict__ G, double* __restrict__ loss_partial) {
    __shared__ double sh[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0;
    if (i < nmask) {
        const int64_t n = (int64_t)mask[i] - index_base;
        const float scale = 2.0f / (float)nmask;
        for (int o = 0; o < O; ++o) {
            const float d = Y[n * L + o] - target[n * O + o];
            e += (double)d * (double)d;
            atomicAdd(&G[n * L + o], scale * d);     // a node listed twice contributes twice, like err[mask]
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// column sums and sums of squares of a [rows][dim] array (online-normaliser accumulation): 8 row groups x 32 columns per pass,
// double accumulators, the row groups are added in fixed order
__global__ __launch_bounds__(256) void k_col_stats(const float* __restrict__ x, int64_t rows, int dim, double* __restrict__ partial) {
    __shared__ double sh[2][8][32];
    const int cidx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * STATS_ROWS;
    const int64_t r1 = r0 + STATS_ROWS < rows ? r0 + STATS_ROWS : rows;
    for (int c0 = 0; c0 < dim; c0 += 32) {
        const int col = c0 + cidx;
        double s = 0.0, q = 0.0;
        if (col < dim)
            for (int64_t r = r0 + rg; r < r1; r += 8) {
                const double v = (double)x[r * dim + col];
                s += v;
                q += v * v;
            }
        sh[0][rg][cidx] = s;
        sh[1][rg][cidx] = q;
        __syncthreads();
        if (rg < 2 && col < dim) {
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) t += sh[rg][g][cidx];
            partial[((size_t)blockIdx.x * 2 + rg) * dim + col] = t;
        }
        __syncthreads();
    }
}

// ================================================================================================
// launch wrappers
// ================================================================================================
// 4 tiles per block; two L x L chunk buffers of dynamic LDS (128 KiB at L = 128: the attribute is raised once per kernel)
template <typename K, typename A>
static hipError_t launch_tiles(K kern, const A& a, int ntiles, int L, hipStream_t s, int wpb = 4, size_t extra_lds = 0) {
    if (ntiles <= 0) return hipSuccess;
    const size_t lds = (size_t)2 * L * L * sizeof(float) + extra_lds;
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> granted;
    {
        std::lock_guard<std::mutex> lk(mu);
        size_t& g = granted[reinterpret_cast<const void*>(kern)];
        if (lds > 48 * 1024 && g < lds) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            g = lds;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((ntiles + wpb - 1) / wpb)), dim3(64 * wpb), lds, s, a);
    return hipGetLastError();
}

// eight tiles per block once a launch fills the chip twice over with four (L = 128; MGN_TRAIN_WPB = 4 / 8 forces)
static bool train_wpb8(int L, int ntiles) {
    static const int forced = [] { const char* e = getenv("MGN_TRAIN_WPB"); return e ? atoi(e) : 0; }();
    if (L != 128) return false;
    if (forced) return forced == 8;
    return ntiles > 2048;
}

// Cooperative tiles while a launch has fewer than MGN_TRAIN_COOP_TILES_PER_CU tiles per CU (default 8), L = 128.
static bool train_coop(int L, int ntiles) {
    static const int per_cu = [] { const char* e = getenv("MGN_TRAIN_COOP_TILES_PER_CU"); return e ? atoi(e) : 8; }();
    return L == 128 && ntiles > 0 && ntiles <= per_cu * 256;
}
template <typename K, typename A>
static hipError_t launch_coop(K kern, const A& a, int ntiles, hipStream_t s) {
    hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(256), 2 * 16 * 64 * sizeof(f32x4), s, a);
    return hipGetLastError();
}

bool train_uses_coop(int L, int ntiles) { return train_coop(L, ntiles); }
bool train_fwd_fused_agg(int L, int ntiles) {
    static const int on = [] { const char* e = getenv("MGN_TRAIN_FUSED_AGG"); return e ? atoi(e) : 1; }();
    return on && L == 128 && !train_coop(L, ntiles) && train_wpb8(L, ntiles);
}
hipError_t launch_seg_fixup(int L, const int32_t* rowptr, const float* carry, float* agg, int32_t n, hipStream_t s) {
    const int64_t tot = (int64_t)n * (L / 4);
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_seg_fixup, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, rowptr, carry, agg, n, L / 4);
    return hipGetLastError();
}
// the streaming kernels at L = 128 compute on two fp16 pieces per operand, three piece products (train_chunk); MGN_TRAIN_F16=0: fp32 MFMA
static int g_train_f16 = [] { const char* e = getenv("MGN_TRAIN_F16"); return e ? atoi(e) : 1; }();
int set_train_f16(int on) { const int old = g_train_f16; g_train_f16 = on; return old; }

hipError_t launch_lin2(int L, const Lin2Args& a, hipStream_t s) {
    if (train_wpb8(L, a.ntiles)) return g_train_f16 ? launch_tiles(k_lin2<4, 8, true>, a, a.ntiles, L, s, 8) : launch_tiles(k_lin2<4, 8>, a, a.ntiles, L, s, 8);
    if (L == 128) return g_train_f16 ? launch_tiles(k_lin2<4, 4, true>, a, a.ntiles, L, s) : launch_tiles(k_lin2<4, 4>, a, a.ntiles, L, s);
    if (L == 64) return launch_tiles(k_lin2<2, 4>, a, a.ntiles, L, s);
    if (L == 32) return launch_tiles(k_lin2<1, 4>, a, a.ntiles, L, s);
    return hipErrorInvalidValue;
}

hipError_t launch_mlp_fwd(int L, int nin, const TrainFwdArgs& a, hipStream_t s) {
    if (train_coop(L, a.ntiles)) {
        if (g_train_f16) {
            if (nin == 1) return launch_coop(k_mlp_fwd_coop<1, true>, a, a.ntiles, s);
            if (nin == 2) return launch_coop(k_mlp_fwd_coop<2, true>, a, a.ntiles, s);
            if (nin == 3) return launch_coop(k_mlp_fwd_coop<3, true>, a, a.ntiles, s);
        }
        if (nin == 1) return launch_coop(k_mlp_fwd_coop<1>, a, a.ntiles, s);
        if (nin == 2) return launch_coop(k_mlp_fwd_coop<2>, a, a.ntiles, s);
        if (nin == 3) return launch_coop(k_mlp_fwd_coop<3>, a, a.ntiles, s);
    }
#define FWD_CASE(NT_, NIN_) if (L == 32 * NT_ && nin == NIN_) return launch_tiles(k_mlp_fwd<NT_, NIN_, 4>, a, a.ntiles, L, s)
    if (train_wpb8(L, a.ntiles)) {
        if (g_train_f16) {
            if (nin == 1) return launch_tiles(k_mlp_fwd<4, 1, 8, true>, a, a.ntiles, L, s, 8);
            if (nin == 2) return launch_tiles(k_mlp_fwd<4, 2, 8, true>, a, a.ntiles, L, s, 8);
            if (nin == 3) return launch_tiles(k_mlp_fwd<4, 3, 8, true>, a, a.ntiles, L, s, 8);
        }
        if (nin == 1) return launch_tiles(k_mlp_fwd<4, 1, 8>, a, a.ntiles, L, s, 8);
        if (nin == 2) return launch_tiles(k_mlp_fwd<4, 2, 8>, a, a.ntiles, L, s, 8);
        if (nin == 3) return launch_tiles(k_mlp_fwd<4, 3, 8>, a, a.ntiles, L, s, 8);
    }
    if (L == 128 && g_train_f16) {
        if (nin == 1) return launch_tiles(k_mlp_fwd<4, 1, 4, true>, a, a.ntiles, L, s);
        if (nin == 2) return launch_tiles(k_mlp_fwd<4, 2, 4, true>, a, a.ntiles, L, s);
        if (nin == 3) return launch_tiles(k_mlp_fwd<4, 3, 4, true>, a, a.ntiles, L, s);
    }
    FWD_CASE(4, 1); FWD_CASE(4, 2); FWD_CASE(4, 3);
    FWD_CASE(2, 1); FWD_CASE(2, 2); FWD_CASE(2, 3);
    FWD_CASE(1, 1); FWD_CASE(1, 2); FWD_CASE(1, 3);
#undef FWD_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_mlp_bwd(int L, int nin, const TrainBwdArgs& a, hipStream_t s) {
    if (train_coop(L, a.ntiles)) {
        if (g_train_f16) {
            if (nin == 1) return launch_coop(k_mlp_bwd_coop<1, true>, a, a.ntiles, s);
            if (nin == 2) return launch_coop(k_mlp_bwd_coop<2, true>, a, a.ntiles, s);
            if (nin == 3) return launch_coop(k_mlp_bwd_coop<3, true>, a, a.ntiles, s);
        }
        if (nin == 1) return launch_coop(k_mlp_bwd_coop<1>, a, a.ntiles, s);
        if (nin == 2) return launch_coop(k_mlp_bwd_coop<2>, a, a.ntiles, s);
        if (nin == 3) return launch_coop(k_mlp_bwd_coop<3>, a, a.ntiles, s);
    }
#define BWD_CASE(NT_, NIN_) if (L == 32 * NT_ && nin == NIN_) return launch_tiles(k_mlp_bwd<NT_, NIN_, 4>, a, a.ntiles, L, s)
    if (train_wpb8(L, a.ntiles)) {
        const size_t xl = a.LNSUM ? (size_t)8 * 2 * L * sizeof(float) : 0;       // the waves' column sums behind the two chunk buffers
        if (g_train_f16) {
            if (nin == 1) return launch_tiles(k_mlp_bwd<4, 1, 8, true>, a, a.ntiles, L, s, 8, xl);
            if (nin == 2) return launch_tiles(k_mlp_bwd<4, 2, 8, true>, a, a.ntiles, L, s, 8, xl);
            if (nin == 3) return launch_tiles(k_mlp_bwd<4, 3, 8, true>, a, a.ntiles, L, s, 8, xl);
        }
        if (nin == 1) return launch_tiles(k_mlp_bwd<4, 1, 8>, a, a.ntiles, L, s, 8, xl);
        if (nin == 2) return launch_tiles(k_mlp_bwd<4, 2, 8>, a, a.ntiles, L, s, 8, xl);
        if (nin == 3) return launch_tiles(k_mlp_bwd<4, 3, 8>, a, a.ntiles, L, s, 8, xl);
    }
    if (L == 128 && g_train_f16) {
        if (nin == 1) return launch_tiles(k_mlp_bwd<4, 1, 4, true>, a, a.ntiles, L, s);
        if (nin == 2) return launch_tiles(k_mlp_bwd<4, 2, 4, true>, a, a.ntiles, L, s);
        if (nin == 3) return launch_tiles(k_mlp_bwd<4, 3, 4, true>, a, a.ntiles, L, s);
    }
    BWD_CASE(4, 1); BWD_CASE(4, 2); BWD_CASE(4, 3);
    BWD_CASE(2, 1); BWD_CASE(2, 2); BWD_CASE(2, 3);
    BWD_CASE(1, 1); BWD_CASE(1, 2); BWD_CASE(1, 3);
#undef BWD_CASE
    return hipErrorInvalidValue;
}

static int64_t wgrad_rows_per_block(int64_t rows) {
    // (A/B on the cylinder mesh, k_wgrad_lds: 64 / 128 rows 3.13 / 3.11 ms per step, 256: 3.68, 512: 4.90 -- a block's row loop is a latency chain;
    // round 6, k_wgrad_h2 (two blocks per CU): 128 / 160 / 192 / 224 / 256 / 384 rows 2.45 / 2.39 / 2.38 / 2.43 / 2.53 / 2.90 ms, k_wgrad_lds 2.44 at 128, 2.48 at 192)
    static const int min_rows = [] {
        const char* e = getenv("MGN_WG_MIN_ROWS");
        if (e) return atoi(e);
        const char* h = getenv("MGN_WGRAD_H2");
        return (h && atoi(h) == 0) ? WG_ROWS : 192;
    }();
    int64_t rpb = (rows + 1023) / 1024;              // at most 1024 blocks
    if (rpb < min_rows) rpb = min_rows;
    return (rpb + 2 * WG_UNROLL - 1) / (2 * WG_UNROLL) * (2 * WG_UNROLL);
}
static bool wgrad_h2_on(int L) {
    static const int h2 = [] { const char* e = getenv("MGN_WGRAD_H2"); return e ? atoi(e) : 1; }();      // 0: the fp32 MFMA forms
    return L == 128 && h2 && g_train_f16;
}
bool train_bwd_fused_sgr(int L, int ntiles) {
    static const int on = [] { const char* e = getenv("MGN_TRAIN_FUSED_SGR"); return e ? atoi(e) : 0; }();   // (built, parity green, same box 0.304 s against 0.302 with k_segment_sum_pair: off)
    return on && L == 128 && !train_coop(L, ntiles) && train_wpb8(L, ntiles);
}
bool train_bwd_ln_sums(int L, int ntiles) {
    static const int on = [] { const char* e = getenv("MGN_TRAIN_BWD_LN_SUMS"); return e ? atoi(e) : 1; }();   // LayerNorm-parameter sums inside the streaming backward kernel (0: the LayerNorm job of the weight-gradient launch)
    return on && L == 128 && !train_coop(L, ntiles) && train_wpb8(L, ntiles);
}
hipError_t launch_colsum_groups(const float* part, int nblocks, int cols, int groups, float* out, hipStream_t s) {
    if (nblocks <= 0 || groups <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_colsum_groups, dim3(groups), dim3(256), 0, s, part, nblocks, cols, groups, out);
    return hipGetLastError();
}
bool wgrad_ln_jobs(int L) {
    static const int on = [] { const char* e = getenv("MGN_WGRAD_LN_JOBS"); return e ? atoi(e) : 1; }();  // 0: GT / G xhat rows written by the backward kernel, two column-sum jobs
    return on && wgrad_h2_on(L);
}
int wgrad_blocks_of_job(int64_t launch_rows, int64_t job_rows) {
    if (launch_rows <= 0 || job_rows <= 0) return 0;
    const int64_t rpb = wgrad_rows_per_block(launch_rows);
    return (int)((job_rows + rpb - 1) / rpb);
}
int wgrad_blocks(int64_t rows) {
    if (rows <= 0) return 0;
    const int64_t rpb = wgrad_rows_per_block(rows);
    return (int)((rows + rpb - 1) / rpb);
}

// the inference layouts from the parameter vector, on the device (mgn_api.cpp: pack_inference_weights)
DEVINL uint16_t pk_bf16(float f) {                      // round to nearest even (finite weights)
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
DEVINL float pk_f32(uint16_t b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__global__ void k_pack_weights(const WPackJob* __restrict__ jobs, const float* __restrict__ params, float* __restrict__ wfrag,
                               uint16_t* __restrict__ wsp, uint16_t* __restrict__ wbf, int L) {
    const WPackJob jb = jobs[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L * L) return;
    const float* W = params + jb.src + (long long)jb.kbase * jb.ldw;
    if (jb.kind == 0) {
        const int NT = L / 32, J = L / 2;
        {   // fragment order + its t-major copy: element (j, lane, t)
            const int t = idx % NT, lane = (idx / NT) % 64, j = idx / (NT * 64);
            const int hh = lane >> 5, i = lane & 31;
            const int row = 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * hh;
            const float v = W[(long long)row * jb.ldw + 32 * t + i];
            wfrag[jb.off + idx] = v;
            wfrag[jb.off + (long long)L * L + (((long long)t * (J / 4) + j / 4) * 64 + lane) * 4 + (j & 3)] = v;
        }
        if (L == 128) {   // 16x16x4 order: dst[(((w * 8 + bb) * 2 + j) * 64 + lane) * 4 + i] = W[16 bb + 4 (lane >> 4) + i][16 (2 w + j) + (lane & 15)]
            const int i = idx & 3, lane = (idx >> 2) & 63, j = (idx >> 8) & 1, bb = (idx >> 9) & 7, w = idx >> 12;
            wfrag[jb.off + 2LL * L * L + idx] = W[(long long)(16 * bb + 4 * (lane >> 4) + i) * jb.ldw + 16 * (2 * w + j) + (lane & 15)];
        }
        return;
    }
    // bf16 layouts (L = 128): element j of lane `lane` of fragment `fr`
    const int j = idx & 7, lane = (idx >> 3) & 63, fr = idx >> 9;
    int k, n;
    if (jb.kind == 2 || jb.kind == 5) {           // [ks][ob][lane][8]: input 16 (2 ks + (j >> 2)) + 4 (lane >> 4) + (j & 3), output 16 ob + (lane & 15)
        const int ks = fr >> 3, ob = fr & 7;
        k = 16 * (2 * ks + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
        n = 16 * ob + (lane & 15);
    } else {                      // [s][t][lane][8]: input 32 (s >> 1) + 16 (s & 1) + 8 (j >> 2) + 4 hh + (j & 3), output 32 t + i
        const int sidx = fr >> 2, t = fr & 3, hh = lane >> 5, i = lane & 31;
        k = 32 * (sidx >> 1) + 16 * (sidx & 1) + 8 * (j >> 2) + 4 * hh + (j & 3);
        n = 32 * t + i;
    }
    const float w = W[(long long)k * jb.ldw + n];
    if (jb.kind == 3) {
        wbf[jb.off + idx] = pk_bf16(w);
        return;
    }
    if (jb.kind == 4 || jb.kind == 5) {                 // w scale = hi + lo (+ <= 2^-23 relative), both fp16 (split_common.hpp)
        const float ws = w * jb.scale;
        const _Float16 hi = (_Float16)ws;
        const _Float16 lo = (_Float16)(ws - (float)hi);
        wsp[jb.off + idx] = __builtin_bit_cast(uint16_t, hi);
        wsp[jb.off + 16384 + idx] = __builtin_bit_cast(uint16_t, lo);
        return;
    }
    const uint16_t hi = pk_bf16(w);                     // w = hi + mid + lo exactly (split.hip)
    const float r1 = w - pk_f32(hi);
    const uint16_t mid = pk_bf16(r1);
    const float r2 = r1 - pk_f32(mid);
    wsp[jb.off + idx] = hi;
    wsp[jb.off + 16384 + idx] = mid;
    wsp[jb.off + 2 * 16384 + idx] = pk_bf16(r2);
}
hipError_t launch_pack_weights(int L, const WPackJob* jobs, int njobs, const float* params, float* wfrag, uint16_t* wsp, uint16_t* wbf, hipStream_t s) {
    if (njobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_weights, dim3((L * L + 255) / 256, njobs), dim3(256), 0, s, jobs, params, wfrag, wsp, wbf, L);
    return hipGetLastError();
}

// the training weights from the parameter vector, on the device (mgn_train.cpp: pack_training_weights; the host twins are pack_chunk /
// pack_chunk_tmajor of mgn_api.cpp): one block column per job
// element (k, n) of the L x L block a PackJob describes (zero-padded, transposed on request; src < 0: the identity)
DEVINL float pack_elem(const PackJob& jb, const float* __restrict__ params, int k, int n) {
    if (jb.src < 0) return k == n ? 1.f : 0.f;
    if (jb.transpose) return (n < jb.nr && k < jb.nc) ? params[jb.src + (long long)(jb.r0 + n) * jb.ldw + k] : 0.f;
    return (k < jb.nr && n < jb.nc) ? params[jb.src + (long long)(jb.r0 + k) * jb.ldw + n] : 0.f;
}
// largest magnitude of every chunk that wants fp16 pieces (scale > 0), as the bits of a non-negative float: the pieces' power of two is
// taken from it ON THE DEVICE (a training loop packs new weights before every step: ~300 host passes over 16 K values were 2.7 ms of it)
__global__ void k_pack_absmax(const PackJob* __restrict__ jobs, const float* __restrict__ params, unsigned* __restrict__ jobmax, int L) {
    const PackJob jb = jobs[blockIdx.y];
    if (jb.kind != 0 || !(jb.scale > 0.f)) return;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    float m = idx < L * L ? __builtin_fabsf(pack_elem(jb, params, idx / L, idx % L)) : 0.f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    // one atomic per block (as one per wave, 256 atomics per chunk on one address: 185 us per parameter change for the 15-step model)
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(jobmax + blockIdx.y, __builtin_bit_cast(unsigned, __builtin_fmaxf(__builtin_fmaxf(wm[0], wm[1]), __builtin_fmaxf(wm[2], wm[3]))));
}
__global__ void k_pack_train(const PackJob* __restrict__ jobs, const float* __restrict__ params, const float* __restrict__ tabs,
                             const unsigned* __restrict__ jobmax, float* __restrict__ out, int L) {
    const PackJob jb = jobs[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (jb.kind == 1) {
        if (idx < T_COUNT * L) out[jb.off + idx] = tabs[jb.src + idx];
        return;
    }
    const int NT = L / 32, J = L / 2;
    if (idx >= L * L) return;
    const int t = idx % NT, lane = (idx / NT) % 64, j = idx / (NT * 64);
    const int hh = lane >> 5, i = lane & 31;
    const int row = 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * hh;      // phi(j, hh): the input feature of k-step j, half hh
    const int col = 32 * t + i;
    float v;
    if (jb.src < 0) v = row == col ? 1.f : 0.f;
    else if (jb.transpose) v = (col < jb.nr && row < jb.nc) ? params[jb.src + (long long)(jb.r0 + col) * jb.ldw + row] : 0.f;
    else v = (row < jb.nr && col < jb.nc) ? params[jb.src + (long long)(jb.r0 + row) * jb.ldw + col] : 0.f;
    out[jb.off + idx] = v;
    out[jb.off + (long long)L * L + (((long long)t * (J / 4) + j / 4) * 64 + lane) * 4 + (j & 3)] = v;
    if (jb.scale > 0.f && L == 128) {   // the same chunk as two fp16 pieces (h2_chunk_inplace): element jj of lane `ln` of fragment (s, tt) = W[k][n]
        const int jj = idx & 7, ln = (idx >> 3) & 63, fr = idx >> 9;
        const int sidx = fr >> 2, tt = fr & 3, hh2 = ln >> 5, i2 = ln & 31;
        const int k = 32 * (sidx >> 1) + 16 * (sidx & 1) + 8 * (jj >> 2) + 4 * hh2 + (jj & 3), n = 32 * tt + i2;
        const H2Scale sc = h2_scale(__builtin_bit_cast(float, jobmax[blockIdx.y]));      // the chunk's largest entry lands in [2^14, 2^15)
        const float ws = pack_elem(jb, params, k, n) * sc.s;
        const _Float16 hi = (_Float16)ws;
        const _Float16 lo = (_Float16)(ws - (float)hi);
        uint16_t* pc = reinterpret_cast<uint16_t*>(out + jb.off + 2LL * L * L);
        pc[idx] = __builtin_bit_cast(uint16_t, hi);
        pc[16384 + idx] = __builtin_bit_cast(uint16_t, lo);
        if (idx == 0) out[jb.off + 3LL * L * L] = sc.rs;
    }
}
hipError_t launch_pack_train(int L, const PackJob* jobs, int njobs, const float* params, const float* tabs, unsigned* jobmax, float* out,
                             hipStream_t s) {
    if (njobs <= 0) return hipSuccess;
    const int n = L * L > T_COUNT * L ? L * L : T_COUNT * L;
    if (hipError_t e = hipMemsetAsync(jobmax, 0, (size_t)njobs * sizeof(unsigned), s)) return e;
    if (L == 128) hipLaunchKernelGGL(k_pack_absmax, dim3((L * L + 255) / 256, njobs), dim3(256), 0, s, jobs, params, jobmax, L);
    hipLaunchKernelGGL(k_pack_train, dim3((n + 255) / 256, njobs), dim3(256), 0, s, jobs, params, tabs, jobmax, out, L);
    return hipGetLastError();
}

hipError_t launch_wgrad(int L, WgradBatch wb, int64_t rows, hipStream_t s) {
    const int nb = wgrad_blocks(rows);
    if (nb == 0 || wb.njobs <= 0) return hipSuccess;
    wb.rows_per_block = wgrad_rows_per_block(rows);
    const dim3 grid(nb, wb.njobs);
    static const int lds = [] { const char* e = getenv("MGN_WGRAD_LDS"); return e ? atoi(e) : 1; }();   // 0: k_wgrad<4> (4-byte operand loads)
    if (wgrad_h2_on(L)) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_h2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WH_LDS);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL(k_wgrad_h2, grid, dim3(256), WH_LDS, s, wb);
    } else if (L == 128 && lds) hipLaunchKernelGGL(k_wgrad_lds, grid, dim3(256), 0, s, wb);
    else if (L == 128) hipLaunchKernelGGL(k_wgrad<4>, grid, dim3(256), 0, s, wb);
    else if (L == 64) hipLaunchKernelGGL(k_wgrad<2>, grid, dim3(128), 0, s, wb);
    else if (L == 32) hipLaunchKernelGGL(k_wgrad<1>, grid, dim3(64), 0, s, wb);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_reduce_partials(const ReduceBatch& rb, hipStream_t s) {
    int n = 0;
    for (int j = 0; j < rb.njobs; ++j) n = rb.job[j].nrows * rb.job[j].cols > n ? rb.job[j].nrows * rb.job[j].cols : n;
    if (n <= 0 || rb.njobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_reduce_partials, dim3((n + 255) / 256, rb.njobs), dim3(256), 0, s, rb);
    return hipGetLastError();
}

hipError_t launch_segment_sum(int L, const float* src, const int32_t* rowptr, const int32_t* perm, const float* add, float* out,
                              int32_t n, hipStream_t s) {
    const int64_t tot = (int64_t)n * (L / 4);
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_segment_sum, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, rowptr, perm, add, out, n, L / 4);
    return hipGetLastError();
}

hipError_t launch_segment_sum2(int L, const float* srcA, const int32_t* rowptrA, const float* srcB, const int32_t* rowptrB, const int32_t* permB,
                               const float* add, float* out, int32_t n, hipStream_t s) {
    const int64_t tot = (int64_t)n * (L / 4);
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_segment_sum2, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, srcA, rowptrA, srcB, rowptrB, permB, add, out, n, L / 4);
    return hipGetLastError();
}

hipError_t launch_segment_sum_pair(int L, const float* src, const int32_t* rowptr_r, const int32_t* rowptr_s, const int32_t* perm_s,
                                   float* out_r, float* out_s, int32_t n, hipStream_t s) {
    const int64_t tot = (int64_t)n * (L / 4);
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_segment_sum_pair, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, rowptr_r, rowptr_s, perm_s, out_r, out_s, n,
                       L / 4);
    return hipGetLastError();
}

hipError_t launch_affine_pad_gather(const float* srcA, int wa, const float* srcB, int wb, const float* scale, const float* shift, const int32_t* gid,
                                    float* dst, int L, int64_t rows, hipStream_t s) {
    const int64_t tot = rows * L;
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_affine_pad, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, srcA, wa, srcB, wb, scale, shift, gid, dst, L, rows);
    return hipGetLastError();
}
hipError_t launch_affine_pad(const float* srcA, int wa, const float* srcB, int wb, const float* scale, const float* shift, float* dst, int L,
                             int64_t rows, hipStream_t s) {
    return launch_affine_pad_gather(srcA, wa, srcB, wb, scale, shift, nullptr, dst, L, rows, s);
}

int array_stats_blocks() { return 1024; }

int train_fwd_stat_slots(int L, int ntiles) { return train_coop(L, ntiles) ? 4 * ntiles : ntiles; }

hipError_t launch_array_stats_final(const double* partial, int64_t slots, int64_t n, float eps_in, float eps_out, float* stats, hipStream_t s) {
    hipLaunchKernelGGL(k_array_stats_final, dim3(1), dim3(1024), 0, s, partial, slots, n, eps_in, eps_out, stats);
    return hipGetLastError();
}

hipError_t launch_array_stats(const float* x, int64_t n, double* partial, float eps_in, float eps_out, float* stats, hipStream_t s) {
    int64_t nb = (n + 256 * 32 - 1) / (256 * 32);      // >= 32 values per thread; at most array_stats_blocks() blocks
    nb = nb < 1 ? 1 : (nb > array_stats_blocks() ? array_stats_blocks() : nb);
    hipLaunchKernelGGL(k_array_stats, dim3((unsigned)nb), dim3(256), 0, s, x, n, partial);
    return launch_array_stats_final(partial, nb, n, eps_in, eps_out, stats, s);
}

hipError_t launch_ln_all_apply(const float* y, const float* stats, const float* gamma, const float* beta, const float* resid, float* out,
                               float* lnout, int64_t n, int L, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ln_all_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, stats, gamma, beta, resid, out, lnout, n, L);
    return hipGetLastError();
}

hipError_t launch_rhs_epilogue(const float* Y, int L, int O, const float* os, const float* osh, const float* mask, const int32_t* gid, float* out,
                               int64_t N, hipStream_t s) {
    const int64_t tot = N * O;
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rhs_epilogue, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Y, L, O, os, osh, mask, gid, out, N);
    return hipGetLastError();
}

hipError_t launch_ln_all_apply_segsum(const float* y, const float* stats, const float* gamma, const float* beta, float* e, const int32_t* rowptr,
                                      float* agg, int32_t n, int L, hipStream_t s) {
    const int64_t tot = (int64_t)n * (L / 4);
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ln_all_apply_segsum, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, y, stats, gamma, beta, e, rowptr, agg, n, L / 4);
    return hipGetLastError();
}

int lnall_bwd_blocks() { return 1024; }

hipError_t launch_lnall_bwd(const float* G0, const float* G1, const int32_t* g1idx, const float* Y, const float* stats, const float* gamma,
                            int64_t rows, int L, double* partial, float* m, float* dgamma, float* dbeta, hipStream_t s) {
    if (L > 128) return hipErrorInvalidValue;
    int64_t nb64 = (rows + 63) / 64;                   // >= 32 rows per half block; at most lnall_bwd_blocks() blocks
    const int nb = (int)(nb64 < 1 ? 1 : (nb64 > lnall_bwd_blocks() ? lnall_bwd_blocks() : nb64));
    const int64_t n = rows * L;
    hipLaunchKernelGGL(k_lnall_bwd_cols, dim3(nb), dim3(256), 0, s, G0, G1, g1idx, Y, stats, rows, L, partial);
    hipLaunchKernelGGL(k_lnall_bwd_cols_final, dim3(1), dim3(1024), 0, s, partial, nb, n, L, gamma, stats, dgamma, dbeta, m);
    return hipGetLastError();
}

hipError_t launch_permute_rows(float* dst, const float* src, const int32_t* gid, int64_t rows, int width, bool scatter, hipStream_t s) {
    const int64_t tot = rows * width;
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, dst, src, gid, rows, width, scatter ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_vjp_seed(const float* Y, int L, int O, const float* lambda, const float* vm, const float* os, const float* osh, float* G,
                           float* dxdt, int64_t N, hipStream_t s) {
    const int64_t tot = N * O;
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_vjp_seed, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, Y, L, O, lambda, vm, os, osh, G, dxdt, N);
    return hipGetLastError();
}

hipError_t launch_extract_cols(const float* src, int L, int O, const float* scale, float* dst, int64_t N, hipStream_t s) {
    const int64_t tot = N * O;
    if (tot <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_extract_cols, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, L, O, scale, dst, N);
    return hipGetLastError();
}

int stats_blocks(int64_t rows) { return rows > 0 ? (int)((rows + STATS_ROWS - 1) / STATS_ROWS) : 0; }

hipError_t launch_col_stats(const float* x, int64_t rows, int dim, double* partial, hipStream_t s) {
    const int nb = stats_blocks(rows);
    if (nb == 0 || dim <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_col_stats, dim3(nb), dim3(256), 0, s, x, rows, dim, partial);
    return hipGetLastError();
}

int loss_blocks(int64_t nmask) { return nmask > 0 ? (int)((nmask + 255) / 256) : 0; }

hipError_t launch_loss(const float* Y, int L, const float* target, int O, const int32_t* mask, int64_t nmask, int32_t index_base,
                       float* G, double* loss_partial, hipStream_t s) {
    const int nb = loss_blocks(nmask);
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_loss, dim3(nb), dim3(256), 0, s, Y, L, target, O, mask, nmask, index_base, G, loss_partial);
    return hipGetLastError();
}

}  // namespace mgn
