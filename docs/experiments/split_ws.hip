// Weight-stationary edge step of the fp32 "exact split" path: the processor's edge update (reference: the
// Processor of GraphNetCore, called at src/solve.jl:200 -- gather, edge MLP, LayerNorm, residual, scatter-add) for large fp32
// launches at L = 128, on v_mfma_f32_32x32x16_bf16 with every fp32 operand split exactly into three bf16 pieces (split.hip).
//
// Why another shape of kernel.  The lane-per-row kernels (k_edge_step, k_edge_split2) give every wave ALL output features of its
// 32 rows, so every wave reads every weight fragment once per tile: 288 KiB of bf16 pieces per tile, of which 160 KiB LDS cannot
// hold.  Stamps (tools/diag_stamps_split.py) show that kernel bound by that L2 stream: with 100+ KiB of weight requests queued per
// CU, every other load of the tile waits thousands of cycles behind them.  Here the weights do not move at all:
//   * a block is 4 waves, one per SIMD, 512 registers each.  Wave t owns output features [32 t, 32 t + 32) of all three layers and
//     keeps the hi and mid pieces of its 32 x 128 weight slices in registers for the whole persistent launch (192 registers, the
//     MFMA A operands); the lo pieces (one use per k-step) sit in LDS (96 KiB).  No weight byte is read from L2 after the prologue.
//   * the activations of a 32-row tile travel between layers through LDS as split pieces: after a layer wave t turns its 32 x 32
//     accumulator block into bf16 hi / mid / lo pieces (ReLU folded in) and publishes them as k-steps 2 t, 2 t + 1 of a 24 KiB
//     image; after a barrier every wave reads the whole image as its B operands.  The element order is the bf16 kernels': a
//     lane's accumulator registers are, as they stand, elements of the next layer's B fragment, so a piece is one ds_write_b128.
//   * one wave per SIMD has nobody to hide its VALU work behind -- except its own matrix instructions: four VALU instructions per
//     v_mfma_f32_32x32x16_bf16 issue in its shadow (tools/overlap_probe.hip).  A block therefore runs TWO tile streams, A and B,
//     one stage apart: in every stage (48 MFMAs = one layer of one stream, then a barrier) the VALU work of the OTHER stream --
//     split + publish, LayerNorm, residual, segmented scan, stores, the next tile's requests -- is cut into eight slices, one per
//     k-step of the chain.  Stage table (cur = tile j of the stream, old = tile j - 1, next = tile j + 1):
//         S0  chain A layer 1 | B: LayerNorm statistics (old) ; A: LayerNorm + residual + store (old) ; B: split e (cur)
//         S1  chain B layer 1 | A: ReLU + split layer 1     ; B: LayerNorm + residual + store (old)
//         S2  chain A layer 2 | B: ReLU + split layer 1     ; A: segmented scan + aggregate stores (old)
//         S3  chain B layer 2 | A: ReLU + split layer 2     ; B: scan (old) ; A: requests for next (P, Q, e) and e again for cur
//         S4  chain A layer 3 | B: ReLU + split layer 2     ; B: requests
//         S5  chain B layer 3 | A: LayerNorm statistics (cur) ; A: split e (next)
//   * LayerNorm over features that live in four waves: each wave reduces its 32 features to (sum, M2 about its own mean), the four
//     pairs are combined after the stage's barrier (Chan et al.: M2 = sum M2_t + 32 sum (mean_t - mean)^2): one exchange, stable.
// Storage, tables, carry rows, results: exactly k_edge_step's (same layouts, same scatter-add by segmented scan).
#include <utility>

#include "kernels.h"
#include "split_common.hpp"

namespace mgn {

template <class F, int... I>
DEVINL void ws_for_(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
DEVINL void ws_for(F&& f) {
    ws_for_(f, std::make_integer_sequence<int, N>{});
}

constexpr int WS_IMG = 3 * 8 * 64;        // u32x4 elements of one activation image: [piece][k-step][lane]
constexpr int WS_PIECE = 2048;            // u32x4 fragments of one weight piece: [s][t][lane]
constexpr int WS_LDS_LO = 0;                                  // byte offsets of the LDS areas
constexpr int WS_LDS_X = 3 * 32768;
constexpr int WS_LDS_ST = WS_LDS_X + 2 * WS_IMG * 16;
constexpr int WS_LDS_TAB = WS_LDS_ST + 2 * 4 * 32 * 8;
constexpr int WS_LDS_BYTES = WS_LDS_TAB + T_COUNT * 128 * 4;

struct WsWeights {          // this wave's A operands: hi and mid pieces of the three layers (order: W1e, W2, W3), k-steps 0..7
    u32x4 h[3][8], m[3][8];
};

struct WsStream {
    f32x16 acc;             // accumulator of the chain in flight: this wave's 32 features x 32 rows
    f32x16 old;             // the previous tile's layer-3 output, until LayerNorm and the scan are done with it
    f32x16 pn, qn;          // next tile: P[s], Q[r] quarters (requested half a stage ahead of their use: their lines were
    f32x16 en;              //   touched -- one dword per 128-byte line -- two stages earlier and wait in L2); its e quarter
    f32x16 er;              // e quarter of the tile whose output is in `old`, read again for the residual
    int touch_pq, touch_e;  // destinations of the touch loads
    EdgeIdx ix, ixo, ixn;   // indices of the current / old / next tile
    EdgeIdx raw2;           // ... and of the tile after next, as loaded (nothing may depend on them for an iteration: no wait)
    int tile, tileo, tilen, tile2;   // tile numbers, clamped into the block's range (loads are always in bounds)
    bool on, ono, onn, on2; // whether those tiles exist (a stream that has run out of tiles computes on, stores nothing)
};

DEVINL void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

DEVINL f32x16 ws_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_wop(a), sp_wop(b), c, 0, 0, 0);
}

// ---- one stage: 48 MFMAs of layer LYR on the image `x`, accumulator `acc`; `pre` runs while the first fragments travel, `fill(S)`
// behind the six MFMAs of k-step S.
#ifndef MGN_WS_DIAG_STAGE
#define MGN_WS_DIAG_STAGE -1      // diagnostic builds: stamp the k-steps of this stage instead of the stage starts
#endif
template <int LYR, class Pre, class Fill>
DEVINL void ws_stage(f32x16& acc, const WsWeights& W, const u32x4* x, const u32x4* wlo, Pre&& pre, Fill&& fill) {
    u32x4 nh = x[0 * 8 * 64], nm = x[1 * 8 * 64], nl = x[2 * 8 * 64], n3 = wlo[0];
    pre();
    __builtin_amdgcn_sched_barrier(0);
    ws_for<8>([&](auto S) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const u32x4 bh = nh, bm = nm, bl = nl, a3 = n3;
        if constexpr (s < 7) {
            nh = x[(0 * 8 + s + 1) * 64];
            nm = x[(1 * 8 + s + 1) * 64];
            nl = x[(2 * 8 + s + 1) * 64];
            n3 = wlo[(s + 1) * 4 * 64];
        }
        acc = ws_mfma(a3, bh, acc);                     // small terms first
        acc = ws_mfma(W.m[LYR][s], bm, acc);
        acc = ws_mfma(W.h[LYR][s], bl, acc);
        acc = ws_mfma(W.m[LYR][s], bh, acc);
        acc = ws_mfma(W.h[LYR][s], bm, acc);
        acc = ws_mfma(W.h[LYR][s], bh, acc);
        fill(S);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// ---- task: split this wave's 16 values per lane into bf16 pieces and publish them as k-steps 2 t, 2 t + 1 of the image.
// Slice P (0..7) handles the pair of registers 2 P, 2 P + 1; the unit's three fragments are stored with its fourth pair.
// Slice D (0..3) handles register pairs D and D + 4 (one of each k-step: two independent dependency chains side by side);
// slices >= 4 are empty.  Both k-steps' fragments are stored with slice 3.
template <bool RELU, int D>
DEVINL void ws_split_slice(const f32x16& v, SpPieces (&pc)[2], u32x4* img, int wave) {
    if constexpr (D < 4) {
        sp_split_pair<RELU>(pc[0].h[D], pc[0].m[D], pc[0].l[D], v[2 * D], v[2 * D + 1]);
        sp_split_pair<RELU>(pc[1].h[D], pc[1].m[D], pc[1].l[D], v[8 + 2 * D], v[8 + 2 * D + 1]);
    }
    if constexpr (D == 3) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            u32x4 h, m, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h[i] = pc[u].h[i];
                m[i] = pc[u].m[i];
                l[i] = pc[u].l[i];
            }
            const int s = 2 * wave + u;
            img[(0 * 8 + s) * 64] = h;
            img[(1 * 8 + s) * 64] = m;
            img[(2 * 8 + s) * 64] = l;
        }
    }
}

// x + x(lane ^ 32) on the VALU (v_permlane32_swap: no LDS round trip, which a lone in-order wave would sit out)
// Inline asm: with both operands the same value, hipcc (ROCm 7.2) hands back ONE register for both results of
// __builtin_amdgcn_permlane32_swap (tools: _pl_probe); the two wait states a VALU write -> v_permlane read needs are in the string.
DEVINL float ws_xhalf_sum(float x) {
    float a = x, b = x;                                  // after the swap: a = [x.lo, x.lo], b = [x.hi, x.hi]
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// ---- task: LayerNorm statistics of this wave's 32 features per row: (sum, M2 about the wave's own mean) -> st[wave][row].
// Four partial sums each: a lone wave per SIMD stalls on every dependent instruction, the MFMAs behind it included.
DEVINL void ws_stats(const f32x16& v, float2* st, int wave, int c) {
    float s0 = v[0] + v[4], s1 = v[1] + v[5], s2 = v[2] + v[6], s3 = v[3] + v[7];
    s0 += v[8]; s1 += v[9]; s2 += v[10]; s3 += v[11];
    s0 += v[12]; s1 += v[13]; s2 += v[14]; s3 += v[15];
    const float s = ws_xhalf_sum((s0 + s1) + (s2 + s3));
    const float mean = s * (1.0f / 32);
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
        const float d0 = v[k] - mean, d1 = v[k + 1] - mean, d2 = v[k + 2] - mean, d3 = v[k + 3] - mean;
        q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
    }
    const float q = ws_xhalf_sum((q0 + q1) + (q2 + q3));
    st[wave * 32 + c] = make_float2(s, q);               // both lane halves hold (and store) the same pair
}

// ---- task: LayerNorm from the four waves' statistics, residual, store of the e quarter.  v: pre-LN in, e' out.
struct WsLn {
    float mean, rstd;
};
DEVINL WsLn ws_ln_combine(const float2* st, int c) {
    const float2 p0 = st[0 * 32 + c], p1 = st[1 * 32 + c], p2 = st[2 * 32 + c], p3 = st[3 * 32 + c];
    const float mean = (p0.x + p1.x + p2.x + p3.x) * (1.0f / 128);
    const float d0 = p0.x * (1.0f / 32) - mean, d1 = p1.x * (1.0f / 32) - mean, d2 = p2.x * (1.0f / 32) - mean, d3 = p3.x * (1.0f / 32) - mean;
    const float m2 = (p0.y + p1.y + p2.y + p3.y) + 32.0f * (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
    WsLn r;
    r.mean = mean;
    r.rstd = __builtin_amdgcn_rsqf(m2 * (1.0f / 128) + LN_EPS);       // v_rsq_f32: 1 ulp
    return r;
}
// four registers (one 16-byte piece g) per slice
template <int G>
DEVINL void ws_ln_slice(f32x16& v, f32x16& e, const WsLn& ln, const float* tb, int wave, int h) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(tb + T_GAMMA * 128)[2 * (4 * wave + G) + h];
    const f32x4 bv = reinterpret_cast<const f32x4*>(tb + T_BETA * 128)[2 * (4 * wave + G) + h];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float y = (v[4 * G + i] - ln.mean) * ln.rstd * gv[i] + bv[i];
        v[4 * G + i] = y;                                  // e'
        e[4 * G + i] += y;                                 // e <- e + e'
    }
}

// ---- task: segmented sum of e' over runs of equal receiver inside the tile (this wave's 16 registers), tails to AGG / CARRY
struct WsScan {
    bool c1, c2, c4, c8, cx, tail, to_carry, sl;
    int r;
};
DEVINL WsScan ws_scan_setup(const EdgeIdx& ix, bool on, int c) {
    WsScan q;
    const bool valid = ix.r >= 0;
    q.r = valid ? ix.r : 0;
    const int reff = valid ? q.r : (-4 - c);
    const int rprev = __shfl_up(reff, 1, 32);
    const int rnext = __shfl_down(reff, 1, 32);
    const bool head = (c == 0) || (reff != rprev);
    const unsigned hm = (unsigned)__ballot(head);
    const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
    const int st_in = max(start, c & 16);
    q.c1 = (c - 1 >= st_in); q.c2 = (c - 2 >= st_in); q.c4 = (c - 4 >= st_in); q.c8 = (c - 8 >= st_in);
    q.cx = (c >= 16) && (start <= 15);
    q.tail = on && valid && ((c == 31) || (reff != rnext));
    const int r_first = __builtin_amdgcn_readfirstlane(reff);
    q.sl = (start == 0) && (ix.r_before == r_first);
    const bool sr = (c == 31) && (ix.r_after == reff);
    q.to_carry = q.sl || sr;
    return q;
}
// the five levels on registers [4 G, 4 G + 4): v += dpp(v) * mask (segmented_scan of tile_common.hpp, four registers at a time)
template <int G>
DEVINL void ws_scan_slice(f32x16& v, const WsScan& q) {
    const float m1 = q.c1 ? 1.f : 0.f, m2 = q.c2 ? 1.f : 0.f, m4 = q.c4 ? 1.f : 0.f, m8 = q.c8 ? 1.f : 0.f, mx = q.cx ? 1.f : 0.f;
#define WS_SCAN_LEVEL(M, CTRL)                                                                                       \
    _Pragma("unroll") for (int k = 4 * G; k < 4 * G + 4; ++k)                                                        \
        asm volatile("v_fmac_f32_dpp %0, %0, %1 " CTRL " bound_ctrl:0" : "+v"(v[k]) : "v"(M));
    asm volatile("s_nop 1" : "+v"(v));      // a DPP read needs two wait states behind the VALU write of its source
    WS_SCAN_LEVEL(m1, "row_shr:1 row_mask:0xf bank_mask:0xf")
    WS_SCAN_LEVEL(m2, "row_shr:2 row_mask:0xf bank_mask:0xf")
    WS_SCAN_LEVEL(m4, "row_shr:4 row_mask:0xf bank_mask:0xf")
    WS_SCAN_LEVEL(m8, "row_shr:8 row_mask:0xf bank_mask:0xf")
    WS_SCAN_LEVEL(mx, "row_bcast:15 row_mask:0xa bank_mask:0xf")
#undef WS_SCAN_LEVEL
}

// requests for a tile: indices.  ws_idx_raw only loads (addresses clamped into the arrays); ws_idx_fix applies the padding /
// boundary rules of load_edge_idx_nb to the loaded values -- an iteration later, so that no wait sits behind the request
DEVINL EdgeIdx ws_idx(const EdgeArgs& a, int tile, int c) { return load_edge_idx_nb(a.snd, a.rcv, a.E, tile, c); }
DEVINL EdgeIdx ws_idx_raw(const EdgeArgs& a, int tile, int c) {
    EdgeIdx ix;
    const int64_t e0 = (int64_t)tile * TILE, eid = e0 + c;
    const int64_t ec = eid < a.E ? eid : a.E - 1;
    ix.s = a.snd[ec];
    ix.r = a.rcv[ec];
    // the two boundary words are wave-uniform: kept in vector registers on purpose (an opaque per-lane zero in the address) --
    // hipcc otherwise moves them to scalar registers with v_readfirstlane right behind the load, i.e. waits for it on the spot
    int z = 0;
    asm volatile("" : "+v"(z));
    ix.r_before = a.rcv[(e0 > 0 ? e0 - 1 : 0) + z];
    ix.r_after = a.rcv[(e0 + TILE < a.E ? e0 + TILE : a.E - 1) + z];
    return ix;
}
DEVINL EdgeIdx ws_idx_fix(const EdgeIdx& raw, const EdgeArgs& a, int tile, int c) {
    EdgeIdx ix;
    const int64_t e0 = (int64_t)tile * TILE;
    const bool valid = e0 + c < a.E;
    ix.s = valid ? raw.s : 0;
    ix.r = valid ? raw.r : -1;
    ix.r_before = (tile > 0) ? raw.r_before : -2;
    ix.r_after = (e0 + TILE < a.E) ? raw.r_after : -3;
    return ix;
}

__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_edge_ws(const EdgeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ws[];
    {   // lo pieces of W1e, W2, W3 and the tables -> LDS
        uint16_t* wl = reinterpret_cast<uint16_t*>(smem_ws + WS_LDS_LO);
        copy_to_lds16(wl, a.split[2] + 2 * 16384, 16384, true);
        copy_to_lds16(wl + 16384, a.split[0] + 2 * 16384, 16384, true);
        copy_to_lds16(wl + 2 * 16384, a.split[1] + 2 * 16384, 16384, true);
        copy_to_lds(reinterpret_cast<float*>(smem_ws + WS_LDS_TAB), a.tabs, T_COUNT * L);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    WsWeights W;
    {
        const u32x4* g[3] = {reinterpret_cast<const u32x4*>(a.split[2]), reinterpret_cast<const u32x4*>(a.split[0]),
                             reinterpret_cast<const u32x4*>(a.split[1])};
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                W.h[l][s] = g[l][(s * 4 + wave) * 64 + lane];
                W.m[l][s] = g[l][WS_PIECE + (s * 4 + wave) * 64 + lane];
            }
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int s = 0; s < 8; ++s) {       // park them in the accumulator half of the register file: the MFMAs read them there
                asm volatile("" : "+a"(W.h[l][s]));
                asm volatile("" : "+a"(W.m[l][s]));
            }
    }
    const u32x4* wlo = reinterpret_cast<const u32x4*>(smem_ws + WS_LDS_LO) + wave * 64 + lane;      // + layer * WS_PIECE + s * 256
    u32x4* xa = reinterpret_cast<u32x4*>(smem_ws + WS_LDS_X) + lane;
    u32x4* xb = xa + WS_IMG;
    float2* sta = reinterpret_cast<float2*>(smem_ws + WS_LDS_ST);
    float2* stb = sta + 4 * 32;
    const float* tb = reinterpret_cast<const float*>(smem_ws + WS_LDS_TAB);

    // this block's tiles: one contiguous range per XCD (blocks b, b + 8, .. share one), swept interleaved by its blocks
    const int xcd = blockIdx.x % NUM_XCD, bi = blockIdx.x / NUM_XCD;
    const int per = (a.ntiles + NUM_XCD - 1) / NUM_XCD;
    const int nb = (gridDim.x - xcd + NUM_XCD - 1) / NUM_XCD;
    const int x1 = a.tile0 + min(xcd * per + per, a.ntiles);
    const int t0 = a.tile0 + xcd * per + bi;
    if (t0 >= x1) return;                                   // (the whole block: no barrier is left waiting)
    const int nmy = (x1 - t0 + nb - 1) / nb;                // tiles of this block: t0 + k nb, k < nmy; stream A even k, B odd k
    const int last = t0 + (nmy - 1) * nb;
    auto tile_of = [&](int k, bool& on) {
        on = k < nmy;
        return on ? t0 + k * nb : last;
    };
    __syncthreads();

    WsStream A, B;
    // prologue: indices of the first two tiles per stream, operands of the first; image of A's first tile
    A.tile = tile_of(0, A.on);
    B.tile = tile_of(1, B.on);
    A.tilen = tile_of(2, A.onn);
    B.tilen = tile_of(3, B.onn);
    A.ix = ws_idx(a, A.tile, c);
    B.ix = ws_idx(a, B.tile, c);
    A.ixn = ws_idx(a, A.tilen, c);
    B.ixn = ws_idx(a, B.tilen, c);
    A.tile2 = tile_of(4, A.on2);
    B.tile2 = tile_of(5, B.on2);
    A.raw2 = ws_idx_raw(a, A.tile2, c);
    B.raw2 = ws_idx_raw(a, B.tile2, c);
    A.ixo = A.ix; B.ixo = B.ix;
    A.tileo = A.tile; B.tileo = B.tile;
    A.ono = false; B.ono = false;
    // touch: one dword of every 128-byte line the later quarter loads will read (P / Q rows: lane half 0 / 1; e: 32 lines)
    auto touch = [&](WsStream& X, const EdgeIdx& ix, int tile) __attribute__((always_inline)) {
        const float* row = h ? a.Q + (int64_t)(ix.r >= 0 ? ix.r : 0) * L : a.P + (int64_t)ix.s * L;
        X.touch_pq = *reinterpret_cast<const int*>(row + 32 * wave);
        X.touch_e = *reinterpret_cast<const int*>(a.Elat + (int64_t)tile * (TILE * L) + (4 * wave) * 256 + c * 32);
    };
    auto load_pq = [&](WsStream& X, const EdgeIdx& ix) __attribute__((always_inline)) {
        asm volatile("" ::"v"(X.touch_pq));
        load_quarter(X.pn, row_ptr(a.P, ix.s, L, h), STRIDE_ROW, wave);
        load_quarter(X.qn, row_ptr(a.Q, ix.r >= 0 ? ix.r : 0, L, h), STRIDE_ROW, wave);
    };
    auto load_e = [&](f32x16& dst, int& tch, int tile) __attribute__((always_inline)) {
        asm volatile("" ::"v"(tch));
        load_quarter(dst, tile_ptr(a.Elat, tile, L, lane), STRIDE_TILE, wave);
    };
    touch(A, A.ix, A.tile);
    touch(B, B.ix, B.tile);
    load_pq(A, A.ix);
    load_e(A.en, A.touch_e, A.tile);
    {
        SpPieces pc[2];
        ws_for<8>([&](auto P) __attribute__((always_inline)) { ws_split_slice<false, decltype(P)::value>(A.en, pc, xa, wave); });
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        A.acc[k] = A.pn[k] + A.qn[k];
        A.old[k] = 0.f;
        B.old[k] = 0.f;
        A.er[k] = 0.f;
        B.er[k] = 0.f;
    }
    load_pq(B, B.ix);
    load_e(B.en, B.touch_e, B.tile);
    ws_barrier();

    const int niter = (nmy + 1) / 2 + 1;                    // + 1: the last tiles' LayerNorm / scan run in the following iteration
    const int lane0 = lane;
    (void)lane0;
#if MGN_WS_DIAG_STAGE >= 0
#define WS_STAMP(k) do {} while (0)
#define WS_KSTAMP(stage, k) do { if constexpr ((stage) == MGN_WS_DIAG_STAGE) STAMP(k); } while (0)
#else
#define WS_STAMP(k) STAMP(k)
#define WS_KSTAMP(stage, k) do {} while (0)
#endif
    WsScan sca = ws_scan_setup(A.ix, false, c), scb = ws_scan_setup(B.ix, false, c);   // run structure of the OLD tiles (none yet)
    const auto no_pre = []() __attribute__((always_inline)) {};
    auto agg_store = [&](const WsStream& X, const WsScan& q) __attribute__((always_inline)) {
        f32x4* dst = q.to_carry ? row_ptr(a.CARRY, (int64_t)2 * X.tileo + (q.sl ? 0 : 1), L, h) : tile_ptr(a.AGG, q.r >> 5, L, 32 * h + (q.r & 31));
        if (q.tail) store_quarter(dst, q.to_carry ? STRIDE_ROW : STRIDE_TILE, wave, X.old);
    };
    // Fill slices: the split tasks use k-steps 0..3 (two pairs each), LayerNorm / scan slices 4..7, the next chain's accumulator
    // (bias quarter from the LDS tables, or P + Q) is set up a stage ahead, every store sits in the last slice.
    for (int j = 0; j < niter; ++j) {
        const int stamp_tile = j;
        (void)stamp_tile;
        SpPieces pa[2], pb[2];
        WsLn lna, lnb;
        WS_STAMP(0);
        // ---------------- S0: chain A layer 1 | B: statistics (old), split e (cur), P + Q (cur); A: LayerNorm + residual + store (old)
        ws_stage<0>(A.acc, W, xa, wlo + 0 * WS_PIECE, no_pre, [&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            WS_KSTAMP(0, s);
            if constexpr (s == 0) ws_stats(B.old, stb, wave, c);
            ws_split_slice<false, s>(B.en, pb, xb, wave);
            if constexpr (s == 2) load_pq(B, B.ix);
            if constexpr (s == 3) lna = ws_ln_combine(sta, c);
            if constexpr (s >= 4) ws_ln_slice<s - 4>(A.old, A.er, lna, tb, wave, h);
            if constexpr (s == 6) {
#pragma unroll
                for (int k = 0; k < 16; ++k) B.acc[k] = B.pn[k] + B.qn[k];
                load_e(B.er, B.touch_e, B.tileo);             // for the residual of B's old tile (S1)
            }
            if constexpr (s == 7) {
                if (A.ono && A.ixo.r >= 0) store_quarter(tile_ptr(a.Elat, A.tileo, L, lane), STRIDE_TILE, wave, A.er);
                touch(A, A.ixn, A.tilen);
            }
        });
        ws_barrier();
        WS_STAMP(1);
        // ---------------- S1: chain B layer 1 | A: ReLU + split layer 1; B: LayerNorm + residual + store (old)
        ws_stage<0>(B.acc, W, xb, wlo + 0 * WS_PIECE, no_pre, [&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            WS_KSTAMP(1, s);
            ws_split_slice<true, s>(A.acc, pa, xa, wave);
            if constexpr (s == 3) lnb = ws_ln_combine(stb, c);
            if constexpr (s >= 4) ws_ln_slice<s - 4>(B.old, B.er, lnb, tb, wave, h);
            if constexpr (s == 5) tab_quarter(A.acc, tb + T_B2 * L, wave, h);
            if constexpr (s == 7) {
                if (B.ono && B.ixo.r >= 0) store_quarter(tile_ptr(a.Elat, B.tileo, L, lane), STRIDE_TILE, wave, B.er);
                touch(B, B.ixn, B.tilen);
            }
        });
        ws_barrier();
        WS_STAMP(2);
        // ---------------- S2: chain A layer 2 | B: ReLU + split layer 1; A: scan + aggregate stores (old)
        ws_stage<1>(A.acc, W, xa, wlo + 1 * WS_PIECE, no_pre, [&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            WS_KSTAMP(2, s);
            ws_split_slice<true, s>(B.acc, pb, xb, wave);
            if constexpr (s >= 4) ws_scan_slice<s - 4>(A.old, sca);
            if constexpr (s == 5) tab_quarter(B.acc, tb + T_B2 * L, wave, h);
            if constexpr (s == 7) agg_store(A, sca);
        });
        ws_barrier();
        WS_STAMP(3);
        // ---------------- S3: chain B layer 2 | A: ReLU + split layer 2; B: scan + aggregate stores (old)
        ws_stage<1>(B.acc, W, xb, wlo + 1 * WS_PIECE, no_pre, [&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            WS_KSTAMP(3, s);
            ws_split_slice<true, s>(A.acc, pa, xa, wave);
            if constexpr (s >= 4) ws_scan_slice<s - 4>(B.old, scb);
            if constexpr (s == 5) tab_quarter(A.acc, tb + T_B3 * L, wave, h);
            if constexpr (s == 7) agg_store(B, scb);
        });
        ws_barrier();
        WS_STAMP(4);
        // ---------------- S4: chain A layer 3 | B: ReLU + split layer 2; requests: e (A next); run structure of the tiles in flight
        ws_stage<2>(A.acc, W, xa, wlo + 2 * WS_PIECE, no_pre, [&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            WS_KSTAMP(4, s);
            ws_split_slice<true, s>(B.acc, pb, xb, wave);
            if constexpr (s == 4) load_e(A.en, A.touch_e, A.tilen);
            if constexpr (s == 5) tab_quarter(B.acc, tb + T_B3 * L, wave, h);
            if constexpr (s == 6) sca = ws_scan_setup(A.ix, A.on, c);     // (their scans run in the next iteration)
            if constexpr (s == 7) scb = ws_scan_setup(B.ix, B.on, c);
        });
        ws_barrier();
        WS_STAMP(5);
        // ---------------- S5: chain B layer 3 | A: statistics (cur), split e (next), P + Q (next); requests: e (B next), e again (A cur)
        ws_stage<2>(B.acc, W, xb, wlo + 2 * WS_PIECE, no_pre, [&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            WS_KSTAMP(5, s);
            if constexpr (s == 0) {
                ws_stats(A.acc, sta, wave, c);
                A.old = A.acc;
            }
            ws_split_slice<false, s>(A.en, pa, xa, wave);
            if constexpr (s == 2) load_pq(A, A.ixn);
            if constexpr (s == 4) load_e(B.en, B.touch_e, B.tilen);
            if constexpr (s == 6) {
#pragma unroll
                for (int k = 0; k < 16; ++k) A.acc[k] = A.pn[k] + A.qn[k];
                load_e(A.er, A.touch_e, A.tile);              // for the residual of this tile (S0 of the next iteration)
            }
        });
        WS_STAMP(6);
        // rotate both streams: cur -> old, next -> cur, and request the indices of the tile after next
        B.old = B.acc;
        A.ixo = A.ix; A.tileo = A.tile; A.ono = A.on;
        B.ixo = B.ix; B.tileo = B.tile; B.ono = B.on;
        A.ix = A.ixn; A.tile = A.tilen; A.on = A.onn;
        B.ix = B.ixn; B.tile = B.tilen; B.on = B.onn;
        A.tilen = A.tile2; A.onn = A.on2; A.ixn = ws_idx_fix(A.raw2, a, A.tile2, c);      // (loaded an iteration ago)
        B.tilen = B.tile2; B.onn = B.on2; B.ixn = ws_idx_fix(B.raw2, a, B.tile2, c);
        A.tile2 = tile_of(2 * (j + 3), A.on2);
        B.tile2 = tile_of(2 * (j + 3) + 1, B.on2);
        A.raw2 = ws_idx_raw(a, A.tile2, c);
        B.raw2 = ws_idx_raw(a, B.tile2, c);
        ws_barrier();
        WS_STAMP(7);
    }
}

hipError_t launch_edge_ws(const EdgeArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_edge_ws), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = a.ntiles < cus ? a.ntiles : cus;
    hipLaunchKernelGGL(k_edge_ws, dim3(blocks), dim3(256), WS_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace mgn
