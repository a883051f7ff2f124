// CDNA4 (gfx950) kernels of the MeshGraphNets Encode-Process-Decode hot path.
//
// Replaces what GraphNetCore.jl/Lux/NNlib execute for `mgn.model(graph, ps, st)` (reference
// src/solve.jl:200): per processor step gather v[:,senders]/v[:,receivers], edge MLP + LayerNorm,
// scatter-add by receiver, node MLP + LayerNorm, residuals.
//
// Design ("lane-per-row"):
//   * One wave owns a tile of 32 rows (edges or nodes).  Lane l = (c = l&31, h = l>>5) owns row c and
//     half of its L features.  The MLP is computed TRANSPOSED on v_mfma_f32_32x32x2_f32:
//         D[feature][row] += W^T[feature][k] * X^T[k][row]
//     so the A operand is a weight fragment (pre-permuted on the host, one 16-byte read per lane per
//     k-step for L=128) and the B operand is ONE VGPR of the lane's own row.  The accumulator of layer
//     i (lane (c,h), register rho -> feature phi(rho,h) = 32(rho>>4) + (rho&3) + 8((rho&15)>>2) + 4h)
//     is used verbatim as the B operand of layer i+1: k-step j consumes register j, the weights are
//     stored in that k order.  Activations never leave registers between layers; no LDS round trip,
//     no barrier; LayerNorm is an in-lane sum + one cross-half exchange; residuals are in-register.
//   * Waves are independent (no __syncthreads after the weight preload), so the two waves resident on
//     each SIMD overlap one wave's gather / LayerNorm / stores with the other's MFMA chain.
//   * fp32 weights of one fused MLP exceed LDS (3 x 64 KiB), so two L x L chunks live in LDS for the
//     whole (persistent) launch and the rest stream from L2 through a small register ring.
//   * Edge-MLP layer 1 is factored: [v_s; v_r; e] W1 = P[s] + Q[r] + e W1e with P = v W1s,
//     Q = v W1r + b1 computed per NODE by the previous node kernel (E/N ~ 6 fewer MFMA flops on the
//     two node-side blocks).  The gather therefore initialises the layer-1 accumulator directly.
//   * Scatter-add: edges are receiver-sorted once per trajectory; a wave does a segmented inclusive
//     scan across its 32 lanes (DPP row shifts), segment tails store whole rows with plain stores;
//     segments that straddle tiles go to per-tile carry rows that the node kernel adds.  No atomics,
//     no zero-fill pass, bitwise reproducible.
#include "frag.hpp"
#include "kernels.h"
#include "tile_common.hpp"
#include "split_common.hpp"

#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace mgn {

// (TileWalk: frag.hpp -- shared with the persistent training kernels)

// ------------------------------------------------------------------------------------------------
// General hidden-layer count (GenMlp, kernels.h).  acc enters with the pre-activation of the MLP's first layer.
//   gen_hidden: ReLU, then every middle layer (Dense + ReLU): acc leaves as the last hidden activation
//   gen_final : acc <- b + W acc for the last Dense (no activation)
// Weights stream from L2 (mfma_chunk<NT, false>); one register copy per layer (64 moves against 16 NT^2 MFMAs).
// ------------------------------------------------------------------------------------------------
template <int NT>
DEVINL void gen_hidden(f32x16 (&acc)[NT], f32x16 (&y)[NT], const GenMlp& g, int lane, int h) {
    constexpr int L = 32 * NT;
    relu_frag<NT>(acc);
    for (int m = 0; m < g.nmid; ++m) {
        tab_frag<NT>(y, g.tabs + m * L, h);
        mfma_chunk<NT, false>(y, acc, g.chunk[m], lane);
        relu_frag<NT>(y);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = y[t];
    }
}
template <int NT>
DEVINL void gen_final(f32x16 (&acc)[NT], f32x16 (&y)[NT], const GenMlp& g, int lane, int h) {
    constexpr int L = 32 * NT;
    tab_frag<NT>(y, g.tabs + g.nmid * L, h);
    mfma_chunk<NT, false>(y, acc, g.chunk[g.nmid], lane);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = y[t];
}

// ================================================================================================
// Processor edge step (K3+K4+K5): gather, edge MLP, LayerNorm, residual, segmented scatter.
// chunk[0]=W2 chunk[1]=W3 chunk[2]=W1[2L:3L];  NRES leading chunks are LDS-resident.
// Elat and AGG are tile-major; P, Q, CARRY row-major.
// ================================================================================================
// Register plan (L = 128: three 64-VGPR arrays, nothing else of that size may be live, the kernel must not
// spill: vmcnt retires in order, so a scratch reload issued behind the epilogue stores waits for all of them):
//   layer 1   acc (init P[s]+Q[r], accumulates)   x = e tile (B operand)           y  free
//   layer 2   acc (B operand)                     x   kept for the residual         y  accumulates
//   layer 3   acc accumulates                     x   kept                          y  (B operand)
//   epilogue  acc = e' (LayerNorm, scan, tails)   x += e', stored                   y  free
//   turnover  acc <- P[s'] + Q[r']                x <- e tile of the next tile
// Rejected by same-box A/B: re-reading the e tile instead of keeping x (+2 %, +3 GB fetch per
// launch), prefetching the next tile's P rows or e tile into the free array during the epilogue (spills, or +-0),
// a deeper weight ring, start stagger, a strict MFMA-pipe token between partner waves (all removed again).
// Steps of the third chunk kept in the LDS left over by two resident chunks at L = 128 (0 when it is fully
// resident anyway): 160 KiB - 2 x 64 KiB - tables - 64 spare bytes = 29 632 B = 28 k-steps of 1 KiB.
#ifndef MGN_EDGE_JR
#define MGN_EDGE_JR 28
#endif
template <int NT, int NRES, bool GEN = false>
__global__ __launch_bounds__(512, 2) void k_edge_step(const EdgeArgs a) {
    constexpr int L = 32 * NT, CH = 16 * NT * 64 * NT;
    constexpr int JR = (NRES == 2 && NT == 4) ? MGN_EDGE_JR : 0;   // partial residency of chunk 2
    constexpr int PART = JR * 64 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
#pragma unroll
    for (int r = 0; r < NRES; ++r) copy_to_lds(smem + r * CH, a.chunk[r], CH);
    if (PART > 0) copy_to_lds(smem + NRES * CH, a.chunk[2], PART);
    float* tb = smem + NRES * CH + PART;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();

    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* w2 = NRES > 0 ? smem : a.chunk[0];
    const float* w3 = NRES > 1 ? smem + CH : a.chunk[1];
    const float* w1 = NRES > 2 ? smem + 2 * CH : a.chunk[2];
    stagger_second_half(wave, a.stagger);

    TileWalk tw(a.ntiles, wave);           // walks the launch's tile range [tile0, tile0 + ntiles)
    tw.tile += a.tile0;
    tw.end += a.tile0;
    if (tw.tile >= tw.end) return;
    f32x16 x[NT], acc[NT], y[NT];
    EdgeIdx ix = load_edge_idx_nb(a.snd, a.rcv, a.E, tw.tile, lane0 & 31);
    {
        const int h0 = lane0 >> 5;
        load_frag<NT>(acc, prow_ptr(a.P, ix.s, L, h0), STRIDE_PROW);
        add_frag<NT>(acc, prow_ptr(a.Q, ix.r >= 0 ? ix.r : 0, L, h0), STRIDE_PROW);
        load_frag<NT>(x, tile_ptr(a.Elat, tw.tile, L, lane0), STRIDE_TILE);
    }
    int stamp_tile = 0;
    (void)stamp_tile;
    for (;; ++stamp_tile) {
        OPAQUE_LANE();
        const int tile = tw.tile;
        const int next = tile + tw.stride;
        const bool has_next = next < tw.end;
        STAMP(0);
        // indices of the NEXT tile are fetched now, ahead of this tile's stores (in-order vmcnt)
        // (the last tile of a wave harmlessly re-fetches itself: no divergent control flow around the loads)
        const int nxt = has_next ? next : tile;
        const EdgeIdx ixn = load_edge_idx_nb(a.snd, a.rcv, a.E, nxt, c);   // branch-free: no wait behind the request
        const bool valid = ix.r >= 0;
        const int r = valid ? ix.r : 0;
        f32x4* etile = tile_ptr(a.Elat, tile, L, lane);
        STAMP(1);

        if constexpr (NRES > 2)
            mfma_chunk<NT, true>(acc, x, w1, lane);            // layer 1 (edge part; P,Q,b1 preloaded)
        else
            mfma_chunk_split<NT, JR>(acc, x, smem + NRES * CH, a.chunk[2], lane);
        STAMP(2);
        if constexpr (GEN) {                                    // hidden_layers != 2: h - 1 middle layers, then the last one
            gen_hidden<NT>(acc, y, a.gen, lane, h);
            gen_final<NT>(acc, y, a.gen, lane, h);
        } else {
            relu_frag<NT>(acc);
            tab_frag<NT>(y, tb + T_B2 * L, h);
            mfma_chunk<NT, (NRES > 0)>(y, acc, w2, lane);          // layer 2
            STAMP(3);
            relu_frag<NT>(y);
            tab_frag<NT>(acc, tb + T_B3 * L, h);
            mfma_chunk<NT, (NRES > 1)>(acc, y, w3, lane);          // layer 3
        }
        STAMP(4);
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_PRIO);                   // memory/VALU phase: win issue arbitration
        // the e tile stays in registers across the three chains (x): re-reading it for the residual cost 3 GB of
        // extra fetch per launch on M-1M and 2 % of kernel time (same-box A/B)
        layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);   // acc = e'
        STAMP(5);
#pragma unroll
        for (int t = 0; t < NT; ++t) x[t] += acc[t];            // e <- e + e'
        if (valid) store_frag<NT>(etile, STRIDE_TILE, x);       // padding rows of the last tile stay zero
        STAMP(6);

        // ---- segmented sum of e' over runs of equal receiver (both halves see the same structure)
        const int reff = valid ? r : (-4 - c);
        const int rprev = __shfl_up(reff, 1, 32);
        const int rnext = __shfl_down(reff, 1, 32);
        const bool head = (c == 0) || (reff != rprev);
        const unsigned hm = (unsigned)__ballot(head);
        const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
        const int st_in = max(start, c & 16);
        const bool c1 = (c - 1 >= st_in), c2 = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
        const bool cx = (c >= 16) && (start <= 15);
        segmented_scan<NT>(acc, c1, c2, c4, c8, cx);
        STAMP(7);
        const bool tail = valid && ((c == 31) || (reff != rnext));
        const int r_first = __builtin_amdgcn_readfirstlane(reff);
        const bool sl = (start == 0) && (ix.r_before == r_first);   // run continues from the previous tile
        const bool sr = (c == 31) && (ix.r_after == reff);          // run continues into the next tile
        const bool to_carry = sl || sr;
        f32x4* dst = to_carry ? prow_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), L, h)
                              : tile_ptr(a.AGG, r >> 5, L, 32 * h + (r & 31));
        if (tail) store_frag<NT>(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, acc);
        if (!has_next) break;
        PHASE_FENCE();
        // turnover: next accumulator init = P[s'] + Q[r'];  x <- e tile of the next tile
        load_frag<NT>(acc, prow_ptr(a.P, ixn.s, L, h), STRIDE_PROW);
        add_frag<NT>(acc, prow_ptr(a.Q, ixn.r >= 0 ? ixn.r : 0, L, h), STRIDE_PROW);
        load_frag<NT>(x, tile_ptr(a.Elat, nxt, L, lane), STRIDE_TILE);
        __builtin_amdgcn_s_setprio(0);
        ix = ixn;
        tw.tile = next;
    }
}

// ================================================================================================
// Processor node step (K6) + projection of next step's P,Q.
// chunk[0]=W2 chunk[1]=W3 chunk[2]=W1[0:L] chunk[3]=W1[L:2L] chunk[4]=WP chunk[5]=WQ
// V and AGG tile-major; CARRY, P, Q row-major.  CARRY row 2*ntiles_e is the all-zero row.
// ================================================================================================

// NAGG = 2: a second edge set's aggregate (AGG2 / CARRY2 / rowptr2) is a further layer-1 input, chunk[6] streamed.
#ifndef MGN_NODE_PAD
#define MGN_NODE_PAD 0
#endif
constexpr bool NODE_PAD = MGN_NODE_PAD != 0;
template <int NT, int NRES, bool PROJECT, int NAGG = 1, bool GEN = false>
__global__ __launch_bounds__(512, 2) void k_node_step(const NodeArgs a) {
    constexpr int L = 32 * NT, CH = 16 * NT * 64 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool fast = a.ntiles <= MGN_FAST_PRELOAD_TILES;             // few tiles per wave: see copy_to_lds_sel
    {
#pragma unroll
        for (int r = 0; r < (NRES < 4 ? NRES : 4); ++r) copy_to_lds_sel(smem + r * CH, a.chunk[r], CH, fast);
    }
    if (PROJECT && NRES > 4) {
#pragma unroll
        for (int r = 4; r < NRES; ++r) copy_to_lds_sel(smem + r * CH, a.chunk[r], CH, fast);
    }
    float* tb = smem + NRES * CH;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();

    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* w2 = NRES > 0 ? smem : a.chunk[0];
    const float* w3 = NRES > 1 ? smem + CH : a.chunk[1];
    const float* w1v = NRES > 2 ? smem + 2 * CH : a.chunk[2];
    const float* w1a = NRES > 3 ? smem + 3 * CH : a.chunk[3];
    const float* wp = NRES > 4 ? smem + 4 * CH : a.chunk[4];
    const float* wq = NRES > 5 ? smem + 5 * CH : a.chunk[5];
    stagger_second_half(wave, a.stagger);

    for (TileWalk tw(a.ntiles, wave, MGN_SPREAD_ROUNDS_NODE); tw.tile < tw.end; tw.tile += tw.stride) {
        OPAQUE_LANE();
        const int tile = tw.tile;
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        f32x4* vtile = tile_ptr(a.V, tile, L, lane);
        f32x16 v[NT], acc[NT], y[NT];
        load_frag<NT>(v, vtile, STRIDE_TILE);

        {
            LOAD_AGGREGATE(NT, y, a.rowptr, a.AGG, a.CARRY, a.zero_row);

            tab_frag<NT>(acc, tb + T_B1 * L, h);
            mfma_chunk<NT, (NRES > 2), NODE_PAD>(acc, v, w1v, lane);     // layer 1, node part
            mfma_chunk<NT, (NRES > 3), NODE_PAD>(acc, y, w1a, lane);     // layer 1, aggregate part
            if constexpr (NAGG > 1) {
                LOAD_AGGREGATE(NT, y, a.rowptr2, a.AGG2, a.CARRY2, a.zero_row2);
                mfma_chunk<NT, false, NODE_PAD>(acc, y, a.chunk[6], lane);   // layer 1, second edge set's aggregate
            }
            if constexpr (GEN) {
                gen_hidden<NT>(acc, y, a.gen, lane, h);
                gen_final<NT>(acc, y, a.gen, lane, h);
            } else {
                relu_frag<NT, NODE_PAD>(acc);
                tab_frag<NT>(y, tb + T_B2 * L, h);
                mfma_chunk<NT, (NRES > 0), NODE_PAD>(y, acc, w2, lane);      // layer 2
                relu_frag<NT, NODE_PAD>(y);
                tab_frag<NT>(acc, tb + T_B3 * L, h);
                mfma_chunk<NT, (NRES > 1), NODE_PAD>(acc, y, w3, lane);      // layer 3
            }
#ifdef MGN_PRIO_NODE
            __builtin_amdgcn_s_setprio(MGN_PRIO_NODE);
#endif
            layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] += acc[t];        // v <- v + v'
            if (valid) store_frag<NT>(vtile, STRIDE_TILE, v);
#ifdef MGN_PRIO_NODE
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        if constexpr (PROJECT) {
            zero_frag<NT>(acc);
            mfma_chunk<NT, (NRES > 4), NODE_PAD>(acc, v, wp, lane);
            if (valid) store_frag<NT>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, acc);
            tab_frag<NT>(y, tb + T_BQ * L, h);
            mfma_chunk<NT, (NRES > 5), NODE_PAD>(y, v, wq, lane);
            if (valid) store_frag<NT>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, y);
        }
    }
}

// ================================================================================================
// P,Q projection alone (both of its chunks LDS-resident: no weight streaming).  chunk[4]=WP chunk[5]=WQ
// ================================================================================================
#ifndef MGN_PROJ_WAVES
#define MGN_PROJ_WAVES 8
#endif
template <int NT, bool RES>
__global__ __launch_bounds__(MGN_PROJ_WAVES * 64, MGN_PROJ_WAVES / 4) void k_project(const NodeArgs a) {
    constexpr int L = 32 * NT, CH = 16 * NT * 64 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (RES) {
        const bool fast = a.ntiles <= MGN_FAST_PRELOAD_TILES;
        copy_to_lds_sel(smem, a.chunk[4], CH, fast);
        copy_to_lds_sel(smem + CH, a.chunk[5], CH, fast);
    }
    float* tb = smem + (RES ? 2 * CH : 0);
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (TileWalk tw(a.ntiles, wave, MGN_SPREAD_ROUNDS_NODE); tw.tile < tw.end; tw.tile += tw.stride) {
        OPAQUE_LANE();
        const int tile = a.tile0 + tw.tile;
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        f32x16 v[NT], acc[NT], y[NT];
        load_frag<NT>(v, tile_ptr(a.V, tile, L, lane), STRIDE_TILE);
        zero_frag<NT>(acc);
        mfma_chunk<NT, RES>(acc, v, RES ? smem : a.chunk[4], lane);
        if (valid) store_frag<NT>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, acc);
        tab_frag<NT>(y, tb + T_BQ * L, h);
        mfma_chunk<NT, RES>(y, v, RES ? smem + CH : a.chunk[5], lane);
        if (valid) store_frag<NT>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, y);
    }
}

// ================================================================================================
// Cooperative-tile kernels for SMALL graphs (L = 128): one 32-row tile per 4-wave block, wave w owns the
// 32-feature block t = w of every layer (64 dependent MFMAs = 4.1 k cycles instead of a 16.4 k-cycle chunk) and
// the waves exchange their output slices through LDS between layers.  A cylinder_flow-sized mesh (374 edge
// tiles, 63 node tiles) then spreads over ~1500 / 250 waves instead of 374 / 63.  Weights stream from L2 in
// t-major fragment order (chunk_t[(t*64 + j)*64 + lane]), one coalesced 256-B load per k-step and wave.
// ================================================================================================
// (CoopRing, coop_prime / coop_chain*, coop_exchange, coop_layer_norm*: frag.hpp -- shared with the training kernels)

// chunk_t[0]=W2 [1]=W3 [2]=W1e  (t-major)
template <bool FENCE>
__global__ __launch_bounds__(256, 2) void k_edge_coop(const EdgeArgs a) {
    constexpr int L = 128;
    const int stamp_tile = 0;
    (void)stamp_tile;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xch0 = reinterpret_cast<f32x4*>(smem);
    f32x4* xch1 = xch0 + 16 * 64;
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tq = wave;   // feature block owned by this wave
    // the first tile's indices travel while the tables are copied (the gathers depend on them: a serial ~2 us otherwise)
    EdgeIdx ix_first = load_edge_idx(a, a.tile0 + blockIdx.x < a.tile0 + a.ntiles ? a.tile0 + (int)blockIdx.x : a.tile0, lane0 & 31);
    // this wave's quarters of the four tables it needs, straight from L2 into registers: no LDS copy, no barrier before the
    // first tile (one tile per block on a small mesh: the prologue is on the critical path)
    f32x16 b2q, b3q, gq, bq;
    {
        const int h0 = lane0 >> 5;
        tab_quarter(b2q, a.tabs + T_B2 * L, tq, h0);
        tab_quarter(b3q, a.tabs + T_B3 * L, tq, h0);
        tab_quarter(gq, a.tabs + T_GAMMA * L, tq, h0);
        tab_quarter(bq, a.tabs + T_BETA * L, tq, h0);
    }
    STAMP(0);
    for (int tile = a.tile0 + blockIdx.x; tile < a.tile0 + a.ntiles; tile += gridDim.x) {
        OPAQUE_LANE();     // keeps the (loop-invariant) weight and table loads inside the tile loop
        const EdgeIdx ix = (tile == a.tile0 + (int)blockIdx.x) ? ix_first : load_edge_idx(a, tile, c);
        const bool valid = ix.r >= 0;
        const int r = valid ? ix.r : 0;
        f32x16 x[4], in[4], acc, xq;
        f32x4* etile = tile_ptr(a.Elat, tile, L, lane);
        load_frag<4>(x, etile, STRIDE_TILE);
        load_quarter(xq, etile, STRIDE_TILE, tq);
        // the gathered rows depend on the indices (a serial ~2 us); the e tile does not: the layer-1 chain starts from zero on
        // the e tile alone and P[s] + Q[r] (which carry b1) are added when it is done
        f32x16 pq, qq;
        load_quarter(pq, prow_ptr(a.P, ix.s, L, h), STRIDE_PROW, tq);
        load_quarter(qq, prow_ptr(a.Q, r, L, h), STRIDE_PROW, tq);
        STAMP(1);
        CoopRing ring2, ring3;
        coop_prime(ring2, a.chunk_t[0] + tq * 4096, lane);                  // layer 2's first fragments, ahead of time
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        coop_chain<FENCE>(acc, x, a.chunk_t[2] + tq * 4096, lane);                 // layer 1 (edge part)
        acc += pq;
        acc += qq;
        STAMP(2);
        relu_quarter(acc);
        coop_prime(ring3, a.chunk_t[1] + tq * 4096, lane);
        coop_exchange(in, acc, xch0, wave, lane);
        STAMP(3);
        acc = b2q;
        coop_chain_primed<FENCE>(acc, in, a.chunk_t[0] + tq * 4096, lane, ring2);  // layer 2
        relu_quarter(acc);
        coop_exchange(in, acc, xch1, wave, lane);
        STAMP(4);
        acc = b3q;
        coop_chain_primed<FENCE>(acc, in, a.chunk_t[1] + tq * 4096, lane, ring3);  // layer 3
        coop_exchange(in, acc, xch0, wave, lane);                           // full pre-LN row (for the statistics)
        STAMP(5);
        coop_layer_norm_reg(acc, in, gq, bq, a.tabs + T_LN * L);                               // acc = this wave's quarter of e'
        xq += acc;
        if (valid) store_quarter(etile, STRIDE_TILE, tq, xq);
        STAMP(6);
        // segmented sum over runs of equal receiver, this wave's 16 registers
        const int reff = valid ? r : (-4 - c);
        const int rprev = __shfl_up(reff, 1, 32);
        const int rnext = __shfl_down(reff, 1, 32);
        const bool head = (c == 0) || (reff != rprev);
        const unsigned hm = (unsigned)__ballot(head);
        const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
        const int st_in = max(start, c & 16);
        const bool c1 = (c - 1 >= st_in), c2 = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
        const bool cx = (c >= 16) && (start <= 15);
        {
            f32x16 grp[1] = {acc};
            segmented_scan<1>(grp, c1, c2, c4, c8, cx);
            acc = grp[0];
        }
        const bool tail = valid && ((c == 31) || (reff != rnext));
        const int r_first = __builtin_amdgcn_readfirstlane(reff);
        const bool sl = (start == 0) && (ix.r_before == r_first);
        const bool sr = (c == 31) && (ix.r_after == reff);
        const bool to_carry = sl || sr;
        f32x4* dst = to_carry ? prow_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), L, h)
                              : tile_ptr(a.AGG, r >> 5, L, 32 * h + (r & 31));
        if (tail) store_quarter(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, tq, acc);
        STAMP(7);
        __syncthreads();   // xch0 is rewritten by the next tile's first exchange
    }
}

// chunk_t[0]=W2 [1]=W3 [2]=W1v [3]=W1a [4]=WP [5]=WQ (t-major).  mode as in NodeArgs.
template <bool TWO_SETS, bool FENCE>
__global__ __launch_bounds__(256, 2) void k_node_coop(const NodeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xch0 = reinterpret_cast<f32x4*>(smem);
    f32x4* xch1 = xch0 + 16 * 64;
    float* tb = smem + 2 * 16 * 64 * 4;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tq = wave;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        OPAQUE_LANE();
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        f32x16 v[4], in[4], acc, vq;
        f32x4* vtile = tile_ptr(a.V, tile, L, lane);
        load_frag<4>(v, vtile, STRIDE_TILE);
        load_quarter(vq, vtile, STRIDE_TILE, tq);
        if (a.mode != 2) {
            LOAD_AGGREGATE(4, in, a.rowptr, a.AGG, a.CARRY, a.zero_row);
            // every chain's first weight fragments are requested one chain ahead (coop_prime)
            CoopRing ra, rb;
            tab_quarter(acc, tb + T_B1 * L, tq, h);
            coop_prime(ra, a.chunk_t[3] + tq * 4096, lane);
            coop_chain<FENCE>(acc, v, a.chunk_t[2] + tq * 4096, lane);             // layer 1, node part
            coop_prime(rb, a.chunk_t[0] + tq * 4096, lane);
            coop_chain_primed<FENCE>(acc, in, a.chunk_t[3] + tq * 4096, lane, ra); // layer 1, aggregate part
            if constexpr (TWO_SETS) {                                       // second edge set's aggregate
                LOAD_AGGREGATE(4, in, a.rowptr2, a.AGG2, a.CARRY2, a.zero_row2);
                coop_chain<FENCE>(acc, in, a.chunk_t[6] + tq * 4096, lane);
            }
            relu_quarter(acc);
            coop_prime(ra, a.chunk_t[1] + tq * 4096, lane);
            coop_exchange(in, acc, xch0, wave, lane);
            tab_quarter(acc, tb + T_B2 * L, tq, h);
            coop_chain_primed<FENCE>(acc, in, a.chunk_t[0] + tq * 4096, lane, rb); // layer 2
            relu_quarter(acc);
            coop_exchange(in, acc, xch1, wave, lane);
            tab_quarter(acc, tb + T_B3 * L, tq, h);
            if (a.mode == 1) coop_prime(rb, a.chunk_t[4] + tq * 4096, lane);
            coop_chain_primed<FENCE>(acc, in, a.chunk_t[1] + tq * 4096, lane, ra); // layer 3
            coop_exchange(in, acc, xch0, wave, lane);
            coop_layer_norm(acc, in, tb + T_GAMMA * L, tb + T_BETA * L, tq, h);
            vq += acc;                                                      // v <- v + v'  (this wave's quarter)
            if (valid) store_quarter(vtile, STRIDE_TILE, tq, vq);
            if (a.mode == 1) {
                coop_prime(ra, a.chunk_t[5] + tq * 4096, lane);
                coop_exchange(v, vq, xch1, wave, lane);                      // full updated row for the projection
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = 0.f;
                coop_chain_primed<FENCE>(acc, v, a.chunk_t[4] + tq * 4096, lane, rb);
                if (valid) store_quarter(prow_ptr(a.P, nn, L, h), STRIDE_PROW, tq, acc);
                tab_quarter(acc, tb + T_BQ * L, tq, h);
                coop_chain_primed<FENCE>(acc, v, a.chunk_t[5] + tq * 4096, lane, ra);
                if (valid) store_quarter(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, tq, acc);
            }
        }
        if (a.mode == 2) {
            CoopRing rq;
            coop_prime(rq, a.chunk_t[5] + tq * 4096, lane);
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            coop_chain<FENCE>(acc, v, a.chunk_t[4] + tq * 4096, lane);
            if (valid) store_quarter(prow_ptr(a.P, nn, L, h), STRIDE_PROW, tq, acc);
            tab_quarter(acc, tb + T_BQ * L, tq, h);
            coop_chain_primed<FENCE>(acc, v, a.chunk_t[5] + tq * 4096, lane, rq);
            if (valid) store_quarter(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, tq, acc);
        }
        __syncthreads();
    }
}

// ================================================================================================
// 16-row cooperative tiles for SMALL graphs (L = 128, fp32, hidden_layers = 2, one edge set): v_mfma_f32_16x16x4_f32.
// A 32-row tile on v_mfma_f32_32x32x2_f32 is, per wave, a chain of 64 MFMAs of 64 cycles per layer (4.1 k cycles of matrix pipe
// whatever the number of rows), and a cylinder_flow-sized mesh has 63 node tiles for 256 CUs: the step is bound by those serial
// chains (six per node tile).  Half the rows per tile on the 16x16x4 shape (32 cycles per MFMA, 64 per chain) halve the chain time
// and double the number of tiles.  Layout: lane l = (n = l & 15: the row, q = l >> 4); a ROW FRAGMENT is 8 float4: x[bb][i] =
// X[row n][16 bb + 4 q + i]; wave w owns the output blocks 2w, 2w+1: acc[j][i] = feature 16 (2w + j) + 4 q + i of row n -- the
// accumulator layout of the instruction -- so an all-gather of the four waves' slices through LDS is the next layer's operand, and
// k-step (bb, i) contracts register (bb, i) of every lane (weights in that order: pack_chunk16).  HBM storage is unchanged
// (32-row tile-major): a half tile addresses rows 16 (ht & 1) + n of tile ht >> 1.  Carry rows are per 16-edge tile here
// (EdgeArgs.c16: the edge and the node kernel of a step agree; the node kernel also runs behind a 32-row edge kernel on mid-size
// meshes, CSH = 5).  SP (the default wherever the split path is on): the same kernels with their chunks on v_mfma_f32_16x16x32_bf16
// and three-way split operands -- "the same kernels on the split path" further down.
// ================================================================================================
constexpr int C16_CH = 128 * 128;          // floats per chunk copy
DEVINL f32x4 c16_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// float4 index of (feature block bb, lane group q) of row `row` of tile `tile` in 32-row tile-major storage
// (feature o = 16 bb + 4 q sits in piece o >> 3, half (o >> 2) & 1 of the 32-row tile: written so that bb is a constant offset)
DEVINL int64_t c16_tile_idx(int64_t tile, int row, int bb, int q) {
    return (tile * 1024 + (q >> 1) * 64 + 32 * (q & 1) + row) + 128 * bb;
}
// ---- latent storage of the 16-row kernels: fp32 (tile-major float4 pieces / row-major rows) or, in bf16 mode (BF), the bf16
// kernels' arrays: a row is 16 pieces of 8 bf16, piece 2 s + h holding features 32 (s >> 1) + 16 (s & 1) + 8 (j >> 2) + 4 h + (j & 3);
// this lane's four features 16 bb + 4 q + (0..3) are elements 4 (q >> 1) .. + 3 of piece (s = bb, h = q & 1): one 8-byte access.
// Tile-major: [tile][s][32 h + row] pieces; row-major: [row][2 s + h] pieces.  Arithmetic stays fp32 (conversion on load / store).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
DEVINL f32x4 c16_unpack4(const uint2 w) {
    const uint32_t a = w.x, b = w.y;
    f32x4 r;
    r[0] = __builtin_bit_cast(float, a << 16);
    r[1] = __builtin_bit_cast(float, a & 0xFFFF0000u);
    r[2] = __builtin_bit_cast(float, b << 16);
    r[3] = __builtin_bit_cast(float, b & 0xFFFF0000u);
    return r;
}
DEVINL uint2 c16_pack4(const f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
    return __builtin_bit_cast(uint2, o);
}
DEVINL int64_t c16_bf_tile_off(int64_t tile, int row, int bb, int q) { return tile * 4096 + (int64_t)(bb * 64 + 32 * (q & 1) + row) * 8 + 4 * (q >> 1); }
// bf16 P / Q / CARRY rows (16 pieces of 16 bytes): the blocks-of-eight layout of frag.hpp's prow_ptr, in 16-byte pieces
DEVINL int64_t bf_prow_piece(int64_t row, int X) {            // piece X = 2 s + h of a row
#if MGN_PROW_BLOCK
    return (row / MGN_PROW_BLOCK) * (int64_t)(MGN_PROW_BLOCK * 16) + (X >> 1) * (2 * MGN_PROW_BLOCK) + (row % MGN_PROW_BLOCK) * 2 + (X & 1);
#else
    return row * 16 + X;
#endif
}
DEVINL int64_t c16_bf_row_off(int64_t r, int bb, int q) { return bf_prow_piece(r, 2 * bb + (q & 1)) * 8 + 4 * (q >> 1); }
template <bool BF>
DEVINL f32x4 c16_ld_tile(const float* base, int64_t tile, int row, int bb, int q) {
    if constexpr (BF) return c16_unpack4(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + c16_bf_tile_off(tile, row, bb, q)));
    else return reinterpret_cast<const f32x4*>(base)[c16_tile_idx(tile, row, bb, q)];
}
template <bool BF>
DEVINL void c16_st_tile(float* base, int64_t tile, int row, int bb, int q, const f32x4 v) {
    if constexpr (BF) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + c16_bf_tile_off(tile, row, bb, q)) = c16_pack4(v);
    else reinterpret_cast<f32x4*>(base)[c16_tile_idx(tile, row, bb, q)] = v;
}
template <bool BF>
DEVINL f32x4 c16_ld_row(const float* base, int64_t r, int bb, int q) {      // row-major [r][128]
    if constexpr (BF) return c16_unpack4(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + c16_bf_row_off(r, bb, q)));
    else return reinterpret_cast<const f32x4*>(base)[prow_f4(r, q + 4 * bb, 128)];
}
template <bool BF>
DEVINL void c16_st_row(float* base, int64_t r, int bb, int q, const f32x4 v) {
    if constexpr (BF) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + c16_bf_row_off(r, bb, q)) = c16_pack4(v);
    else reinterpret_cast<f32x4*>(base)[prow_f4(r, q + 4 * bb, 128)] = v;
}
// the value a bf16 store keeps (so that what is computed from a row equals what a later launch computes from the stored row)
template <bool BF> DEVINL f32x4 c16_round(const f32x4 v) {
    if constexpr (BF) return c16_unpack4(c16_pack4(v));
    else return v;
}

// one L x L chunk: this wave's two output blocks; wt = chunk16 + w * 4096 floats ([bb][j][lane][4]).  The weight fragments come
// from L2 (a launch of one or two tiles per CU cannot amortise an LDS preload) through a register ring C16_PF k-groups deep, pinned
// by scheduling fences (hipcc otherwise sinks every request to just before its use), and a chain's first fragments can be requested
// ahead of time (c16_prime) -- before the previous chain or the exchange barrier.
#ifndef C16_PF
#define C16_PF 3
#endif
struct C16Ring { f32x4 r[2 * C16_PF]; };
DEVINL void c16_prime(C16Ring& g, const float* wt, int lane) {
    const f32x4* wv = reinterpret_cast<const f32x4*>(wt) + lane;
#pragma unroll
    for (int p = 0; p < 2 * C16_PF; ++p) g.r[p] = wv[p * 64];
}
DEVINL void c16_tab(f32x4 (&acc)[2], const float* tab, int wave, int q) {      // natural-order table -> this wave's slice
    const f32x4* t4 = reinterpret_cast<const f32x4*>(tab) + q;
    acc[0] = t4[4 * (2 * wave)];
    acc[1] = t4[4 * (2 * wave + 1)];
}
DEVINL void c16_relu(f32x4 (&acc)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = fmaxf(acc[j][i], 0.f);
}
// ---- RT consecutive 16-row tiles per block -----------------------------------------------------------------------------------
// A wave keeps RT row tiles, so every weight fragment it fetches feeds 2 RT independent MFMAs.  Why not one tile per block and RT
// blocks per CU (the first version of these kernels; docs/experiments.md, tools/diag_stamps16.py): the second and third block of a
// CU could not even issue their first loads before 2-3 / 6-9 us (each wave streams its own copy of the weights, 48 KB per wave
// and tile through a 64 B / clk texture path), and the co-resident tiles' MFMA and non-MFMA phases did not overlap (12 us without
// any MFMA + 8 us of MFMAs = the 20 us measured).  One block per CU, a third of the weight traffic at RT = 3, no dependent MFMA
// pairs, and the exchange barriers are shared by the RT tiles.
template <int RT, int B0 = 0, int B1 = 8>          // k-groups [B0, B1) of the chunk (the ring carries over between the parts)
DEVINL void c16m_chain(f32x4 (&acc)[RT][2], const f32x4 (&x)[RT][8], const float* wt, int lane, C16Ring& g) {
    const f32x4* wv = reinterpret_cast<const f32x4*>(wt) + lane;
#pragma unroll
    for (int bb = B0; bb < B1; ++bb) {
        const f32x4 c0 = g.r[(2 * bb) % (2 * C16_PF)], c1 = g.r[(2 * bb + 1) % (2 * C16_PF)];
        if (bb + C16_PF < 8) {
            g.r[(2 * bb) % (2 * C16_PF)] = wv[(2 * (bb + C16_PF)) * 64];
            g.r[(2 * bb + 1) % (2 * C16_PF)] = wv[(2 * (bb + C16_PF) + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                acc[t][0] = c16_mfma(c0[i], x[t][bb][i], acc[t][0]);
                acc[t][1] = c16_mfma(c1[i], x[t][bb][i], acc[t][1]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int RT>
DEVINL void c16m_exchange(f32x4 (&full)[RT][8], const f32x4 (&mine)[RT][2], f32x4* xch, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        xch[(t * 8 + 2 * wave) * 64 + lane] = mine[t][0];
        xch[(t * 8 + 2 * wave + 1) * 64 + lane] = mine[t][1];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) full[t][bb] = xch[(t * 8 + bb) * 64 + lane];
}
// LayerNorm of RT tiles whose rows are spread over the four waves (each holds 32 of a row's 128 features: mine[t][j][i]): two passes
// (sum, then sum of squared deviations), each a 32-feature partial per wave combined through red[t][wave][row] in LDS -- instead of
// a third all-gather of the full rows and 4 x redundant statistics.
template <int RT>
DEVINL void c16m_layer_norm(f32x4 (&mine)[RT][2], float* red, const f32x4 (&g)[2], const f32x4 (&b)[2], int wave, int n, const float* lnp) {
    float mean[RT], rstd[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) s += mine[t][j][i];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        red[(t * 4 + wave) * 16 + n] = s;           // the four lane groups write the same value
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RT; ++t)
        mean[t] = ((red[(t * 4 + 0) * 16 + n] + red[(t * 4 + 1) * 16 + n]) + (red[(t * 4 + 2) * 16 + n] + red[(t * 4 + 3) * 16 + n])) * (1.0f / 128);
    float* red2 = red + RT * 64;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = mine[t][j][i] - mean[t];
                v += d * d;
            }
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        red2[(t * 4 + wave) * 16 + n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const float v = (red2[(t * 4 + 0) * 16 + n] + red2[(t * 4 + 1) * 16 + n]) + (red2[(t * 4 + 2) * 16 + n] + red2[(t * 4 + 3) * 16 + n]);
        rstd[t] = ln_rstd_at(v * (1.0f / 128), lnp);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) mine[t][j][i] = (mine[t][j][i] - mean[t]) * rstd[t] * g[j][i] + b[j][i];
    }
}

// ---- the same kernels on the split path (SP): the L x L chunks on v_mfma_f32_16x16x32_bf16, every fp32 operand as three bf16 pieces,
// six piece products (split.hip) -- 48 RT MFMAs of 16 cycles per chunk and wave instead of 64 RT of 32.  The accumulator layout is the
// fp32 kernels' (lane (n, q): features 16 ob + 4 q + i of row n), and k-step ks of the next layer takes blocks 2 ks, 2 ks + 1 -- exactly
// the two blocks wave ks owns: a wave splits ITS slice (ReLU folded in) and the four waves exchange PIECES through LDS (12 KiB per tile
// instead of 8, no redundant split).  Weight pieces: EdgeArgs / NodeArgs::split16 (mgn_api.cpp: pack_chunk16_bf16, [ks][ob][lane][8 bf16]).
// SP = 2 (round 5, the default where the split path is on): TWO fp16 pieces and three piece products (split_common.hpp) -- 24 RT MFMAs per chunk
// and wave.  A row's 128 features are spread over the four waves here, so the power-of-two scale is per (row, k-step): the wave that
// owns a k-step's 32 features takes their maximum (two lane swaps), publishes 1 / scale beside the pieces, and the consumer un-scales every
// k-step's partial sum as it adds it -- acc = fma(partial, rs_row x rs_chunk, acc): the accumulators stay in true units, so bias
// initialisation, LayerNorm and everything else around the chains is the three-piece kernels' code.
template <int NP> struct C16X { u32x4 p[NP]; float rs; };      // one k-step's B operand of a 16-row tile: its pieces (hi first), 1 / its row scale (NP = 2)
typedef C16X<3> C16P;
// largest magnitude (RELU: largest value, at least 0) of a row's 32 features held by this wave: 8 values per lane, lanes n, n + 16, n + 32, n + 48
template <bool RELU>
DEVINL float c16h_rowmax(const f32x4 (&mine)[2]) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const float a = RELU ? mine[j][i] : __builtin_fabsf(mine[j][i]), b = RELU ? mine[j][i + 1] : __builtin_fabsf(mine[j][i + 1]);
            m = __builtin_fmaxf(m, __builtin_fmaxf(a, b));
        }
    m = __builtin_fmaxf(m, __shfl_xor(m, 16, 64));
    return __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
}
template <bool RELU, int NP>
DEVINL C16X<NP> c16s_split(const f32x4 (&mine)[2]) {
    C16X<NP> p;
    if constexpr (NP == 3) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned h, m, l;
            sp_split_pair<RELU>(h, m, l, mine[u >> 1][2 * (u & 1)], mine[u >> 1][2 * (u & 1) + 1]);
            p.p[0][u] = h; p.p[1][u] = m; p.p[2][u] = l;
        }
        p.rs = 1.f;
    } else {
        const H2Scale sc = h2_scale(c16h_rowmax<RELU>(mine));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned h, l;
            h2_split_pair<RELU ? 1 : 0>(h, l, mine[u >> 1][2 * (u & 1)], mine[u >> 1][2 * (u & 1) + 1], sc.s);
            p.p[0][u] = h; p.p[1][u] = l;
        }
        p.rs = sc.rs;
    }
    return p;
}
// exchange buffer of RT tiles: [tile][k-step = wave][piece][lane] 16-byte slots, then (NP = 2) [tile][k-step][row] floats of 1 / scale
template <int RT, int NP> DEVINL float* c16s_scales(u32x4* xch) { return reinterpret_cast<float*>(xch + RT * 4 * NP * 64); }
template <int RT, int NP> DEVINL const float* c16s_scales(const u32x4* xch) { return reinterpret_cast<const float*>(xch + RT * 4 * NP * 64); }
template <int RT, bool RELU, int NP>
DEVINL void c16s_publish(const f32x4 (&mine)[RT][2], u32x4* xch, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const C16X<NP> p = c16s_split<RELU, NP>(mine[t]);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) xch[((t * 4 + wave) * NP + pc) * 64 + lane] = p.p[pc];
        if constexpr (NP == 2) c16s_scales<RT, NP>(xch)[(t * 4 + wave) * 16 + (lane & 15)] = p.rs;      // (the four lane groups write the same value)
    }
    __syncthreads();
}
template <int RT, int NP>
DEVINL void c16s_fetch(C16X<NP> (&x)[RT], const u32x4* xch, int ks, int lane) {      // the pieces of k-step ks of every tile
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) x[t].p[pc] = xch[((t * 4 + ks) * NP + pc) * 64 + lane];
        x[t].rs = NP == 2 ? c16s_scales<RT, NP>(xch)[(t * 4 + ks) * 16 + (lane & 15)] : 1.f;
    }
}
template <int RT, bool RELU, int NP>
DEVINL void c16s_exchange(C16X<NP> (&full)[RT][4], const f32x4 (&mine)[RT][2], u32x4* xch, int wave, int lane) {
    c16s_publish<RT, RELU, NP>(mine, xch, wave, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        C16X<NP> x[RT];
        c16s_fetch<RT, NP>(x, xch, ks, lane);
#pragma unroll
        for (int t = 0; t < RT; ++t) full[t][ks] = x[t];
    }
}
// the weight pieces of a chunk stream from L2 through a register ring C16S_PF (k-step, block) steps deep; step s = 2 ks + j feeds
// this wave's output block 2 wave + j
#ifndef C16S_PF
#define C16S_PF 3
#endif
template <int NP> struct C16SRingT { u32x4 r[NP * C16S_PF]; };
DEVINL const u32x4* c16s_w(const uint16_t* chunk, int wave, int lane) { return reinterpret_cast<const u32x4*>(chunk) + (2 * wave) * 64 + lane; }
template <int NP>
DEVINL void c16s_prime(C16SRingT<NP>& g, const u32x4* wv) {
#pragma unroll
    for (int s = 0; s < C16S_PF; ++s)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) g.r[NP * s + pc] = wv[pc * 2048 + ((s >> 1) * 8 + (s & 1)) * 64];
}
// one (k-step, block) step of RT tiles: six bf16 products into the accumulator, or three fp16 products into a partial sum (zero-based) that
// the caller un-scales as it adds it -- ONE STEP LATER (c16s_apply), behind the next step's MFMAs: a VALU instruction that reads an MFMA
// result waits for the matrix pipe to drain it
template <int RT, int NP>
DEVINL void c16s_step(f32x4 (&acc)[RT][2], f32x4 (&part)[RT], const C16X<NP> (&x)[RT], int j, const u32x4 (&a)[NP]) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        if constexpr (NP == 3) {
            const sp_bf16x8 bh = sp_wop(x[t].p[0]), bm = sp_wop(x[t].p[1]), bl = sp_wop(x[t].p[2]);
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a[2]), bh, acc[t][j], 0, 0, 0);      // small terms first
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a[1]), bm, acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a[0]), bl, acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a[1]), bh, acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a[0]), bm, acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a[0]), bh, acc[t][j], 0, 0, 0);
        } else {
            const sp_f16x8 bh = h2_wop(x[t].p[0]), bl = h2_wop(x[t].p[1]);
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2_wop(a[1]), bh, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2_wop(a[0]), bl, z, 0, 0, 0);
            part[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2_wop(a[0]), bh, z, 0, 0, 0);
        }
    }
}
template <int RT>
DEVINL void c16s_apply(f32x4 (&acc)[RT][2], const f32x4 (&part)[RT], const float (&c)[RT], int j) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][j][i] = __builtin_fmaf(part[t][i], c[t], acc[t][j][i]);
}
template <int RT, int S0 = 0, int S1 = 8, int NP = 3>
DEVINL void c16s_chain(f32x4 (&acc)[RT][2], const C16X<NP> (&x)[RT][4], const u32x4* wv, C16SRingT<NP>& g, float rsw = 1.f) {
    f32x4 part[2][RT];
    float cprev[RT];
#pragma unroll
    for (int s = S0; s < S1; ++s) {
        u32x4 a[NP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) a[pc] = g.r[NP * (s % C16S_PF) + pc];
        if (s + C16S_PF < 8) {
            const int sn = s + C16S_PF;
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) g.r[NP * (s % C16S_PF) + pc] = wv[pc * 2048 + ((sn >> 1) * 8 + (sn & 1)) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        const int ks = s >> 1, j = s & 1;
        C16X<NP> xk[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) xk[t] = x[t][ks];
        c16s_step<RT, NP>(acc, part[s & 1], xk, j, a);
        if constexpr (NP == 2) {
            if (s > S0) c16s_apply<RT>(acc, part[(s - 1) & 1], cprev, (s - 1) & 1);
#pragma unroll
            for (int t = 0; t < RT; ++t) cprev[t] = xk[t].rs * rsw;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NP == 2) c16s_apply<RT>(acc, part[(S1 - 1) & 1], cprev, (S1 - 1) & 1);
}

// four to six row tiles per block: the pieces of all tiles do not fit the registers (48 per tile) -- the exchange only publishes, and the
// chain reads the pieces of ONE k-step at a time from the exchange buffer (12 registers per tile)
template <int RT, int S0 = 0, int S1 = 8, int NP = 3>
DEVINL void c16s_chain_lds(f32x4 (&acc)[RT][2], const u32x4* xch, int lane, const u32x4* wv, C16SRingT<NP>& g, float rsw = 1.f) {
    C16X<NP> x[RT];
    f32x4 part[2][RT];
    float cprev[RT];
#pragma unroll
    for (int s = S0; s < S1; ++s) {
        const int ks = s >> 1, j = s & 1;
        if (j == 0 || s == S0) c16s_fetch<RT, NP>(x, xch, ks, lane);
        u32x4 a[NP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) a[pc] = g.r[NP * (s % C16S_PF) + pc];
        if (s + C16S_PF < 8) {
            const int sn = s + C16S_PF;
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) g.r[NP * (s % C16S_PF) + pc] = wv[pc * 2048 + ((sn >> 1) * 8 + (sn & 1)) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        c16s_step<RT, NP>(acc, part[s & 1], x, j, a);
        if constexpr (NP == 2) {
            if (s > S0) c16s_apply<RT>(acc, part[(s - 1) & 1], cprev, (s - 1) & 1);
#pragma unroll
            for (int t = 0; t < RT; ++t) cprev[t] = x[t].rs * rsw;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NP == 2) c16s_apply<RT>(acc, part[(S1 - 1) & 1], cprev, (S1 - 1) & 1);
}
template <int RT, bool BF, int SP = 0>      // SP: 0 fp32 MFMA pipe, 1 three bf16 pieces, 2 two fp16 pieces
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_edge_coop16m(const EdgeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NP = SP == 2 ? 2 : 3;                             // pieces per operand on the split path
    constexpr int XCH = SP ? RT * 4 * NP * 64 + (SP == 2 ? RT * 16 : 0) : RT * 8 * 64;   // 16-byte slots of one exchange buffer (SP: pieces, then 1 / scale per (tile, k-step, row))
    f32x4* xch0 = reinterpret_cast<f32x4*>(smem);
    f32x4* xch1 = xch0 + XCH;
    float* red = reinterpret_cast<float*>(xch1 + XCH);              // LayerNorm partials: 2 x [RT][4 waves][16 rows]
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* tabs = a.tabs + T_COUNT * L;                       // natural feature order
    const float* w1 = a.chunk[2] + 2 * C16_CH + wave * 4096;
    const float* w2 = a.chunk[0] + 2 * C16_CH + wave * 4096;
    const float* w3 = a.chunk[1] + 2 * C16_CH + wave * 4096;

    const int ht0 = 2 * a.tile0, nht = 2 * a.ntiles;
    const int ngroups = (nht + RT - 1) / RT;
    for (int gi = blockIdx.x; gi < ngroups; gi += gridDim.x) {
        STAMP16(0);
        int lane = lane0;
        asm volatile("" : "+v"(lane));      // keeps the (loop-invariant) weight and table loads inside the loop: hoisted, they spill
        const int n = lane & 15, q = lane >> 4;
        const uint16_t* const* sp16 = SP == 2 ? a.split16h : a.split16;
        const u32x4* s1 = SP ? c16s_w(sp16[2], wave, lane) : nullptr;      // (lane folded in: the ring requests are this + constants)
        const u32x4* s2 = SP ? c16s_w(sp16[0], wave, lane) : nullptr;
        const u32x4* s3 = SP ? c16s_w(sp16[1], wave, lane) : nullptr;
        const float rw1 = a.h2_rs[2], rw2 = a.h2_rs[0], rw3 = a.h2_rs[1];      // (SP = 2) 1 / the chunks' scales
        if ((int64_t)(ht0 + gi * RT) * 16 >= a.E) break;            // nothing but the empty tail of the last 32-row tile
        int ht[RT], s_[RT], r_[RT], r_before[RT], r_after[RT], row[RT];
        int64_t tile[RT];
        bool valid[RT], hb[RT], ha[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            ht[t] = ht0 + gi * RT + t;
            const int64_t e0 = (int64_t)ht[t] * 16;
            const bool live = (gi * RT + t < nht) && e0 < a.E;      // block-uniform; a dead tile computes on clamped rows, stores nothing
            const int64_t eid = e0 + n;
            valid[t] = live && eid < a.E;
            const int64_t ec = valid[t] ? eid : a.E - 1;
            s_[t] = a.snd[ec];
            r_[t] = a.rcv[ec];
            // branch-free (clamped address, select on the value): a conditional load is a branch with a full wait behind it
            // (the selects happen at the use site, behind the chains: placed here they would wait for the loads before layer 1)
            hb[t] = live && ht[t] > 0;
            ha[t] = e0 + 16 < a.E;
            r_before[t] = a.rcv[hb[t] ? e0 - 1 : 0];
            r_after[t] = a.rcv[ha[t] ? e0 + 16 : 0];
            const int htc = live ? ht[t] : ht0 + gi * RT;
            tile[t] = htc >> 1;
            row[t] = 16 * (htc & 1) + n;
        }
        f32x4 x[RT][8], acc[RT][2], xs[RT][2];
        // Few requests before layer 1 (a wave cannot keep sixty cold loads in flight: its first MFMA then waits ~2 us): every wave
        // fetches only its own two blocks of the e rows -- the residual's slice -- and the four waves assemble the full rows through
        // LDS; tables, the next layer's weights and the gathered rows are requested inside the layer-1 chain.
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const float* const esrc = a.ElatSrc ? a.ElatSrc : a.Elat;
            xs[t][0] = c16_ld_tile<BF>(esrc, tile[t], row[t], 2 * wave, q);
            xs[t][1] = c16_ld_tile<BF>(esrc, tile[t], row[t], 2 * wave + 1, q);
        }
        C16Ring g1, g2;
        C16SRingT<NP> h1, h2;
        constexpr bool XL = SP && RT > 3;                             // pieces stay in the exchange buffer (c16s_chain_lds)
        C16X<NP> xp[XL ? 1 : RT][4];
        u32x4* const xs0 = reinterpret_cast<u32x4*>(xch0);
        u32x4* const xs1 = reinterpret_cast<u32x4*>(xch1);
        if constexpr (SP) c16s_prime(h1, s1);
        else c16_prime(g1, w1, lane);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (XL) c16s_publish<RT, false, NP>(xs, xs0, wave, lane);
        else if constexpr (SP) c16s_exchange<RT, false, NP>(xp, xs, xs0, wave, lane);
        else c16m_exchange<RT>(x, xs, xch0, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        STAMP16(1);
        // layer 1: the e tile's part starts as soon as the tile and the first weights are here; the gathered P[s] + Q[r] (which
        // carry b1) are requested half way -- their addresses wait for the index loads, a serial round trip -- and added at the end
        if constexpr (XL) c16s_chain_lds<RT, 0, 4, NP>(acc, xs0, lane, s1, h1, rw1);
        else if constexpr (SP) c16s_chain<RT, 0, 4, NP>(acc, xp, s1, h1, rw1);
        else c16m_chain<RT, 0, 4>(acc, x, w1, lane, g1);
        f32x4 tb2[2], tb3[2], tg[2], tb[2];
        c16_tab(tb2, tabs + T_B2 * L, wave, q);
        c16_tab(tb3, tabs + T_B3 * L, wave, q);
        c16_tab(tg, tabs + T_GAMMA * L, wave, q);
        c16_tab(tb, tabs + T_BETA * L, wave, q);
        if constexpr (SP) c16s_prime(h2, s2);
        else c16_prime(g2, w2, lane);                                // layer 2's first fragments
        f32x4 pq[RT][2][2];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            pq[t][0][0] = c16_ld_row<BF>(a.P, s_[t], 2 * wave, q);
            pq[t][0][1] = c16_ld_row<BF>(a.P, s_[t], 2 * wave + 1, q);
            pq[t][1][0] = c16_ld_row<BF>(a.Q, r_[t], 2 * wave, q);
            pq[t][1][1] = c16_ld_row<BF>(a.Q, r_[t], 2 * wave + 1, q);
        }
        // the index part of the segmented sum (run heads, scan masks, where a run's tail goes) while the gathered rows are on their way:
        // two LDS round trips and a ballot per tile that would otherwise sit between the LayerNorm and the last stores
        float sm1[RT], sm2[RT], sm4[RT], sm8[RT];
        bool tailf[RT], to_carry[RT], sl_[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int reff = valid[t] ? r_[t] : (-4 - n);
            const int rprev = __shfl_up(reff, 1, 16);
            const int rnext = __shfl_down(reff, 1, 16);
            const bool head = (n == 0) || (reff != rprev);
            const unsigned hm = (unsigned)(__ballot(head) & 0xFFFFull);
            const int start = 31 - __clz((int)(hm & (0xFFFFu >> (15 - n))));
            sm1[t] = (n - 1 >= start) ? 1.f : 0.f;
            sm2[t] = (n - 2 >= start) ? 1.f : 0.f;
            sm4[t] = (n - 4 >= start) ? 1.f : 0.f;
            sm8[t] = (n - 8 >= start) ? 1.f : 0.f;
            tailf[t] = valid[t] && ((n == 15) || (reff != rnext));
            const int r_first = __builtin_amdgcn_readfirstlane(reff);
            sl_[t] = (start == 0) && hb[t] && (r_before[t] == r_first);           // run continues from the previous 16-edge tile
            const bool sr = (n == 15) && ha[t] && (r_after[t] == reff);          // run continues into the next one
            to_carry[t] = sl_[t] || sr;
        }
        if constexpr (XL) c16s_chain_lds<RT, 4, 8, NP>(acc, xs0, lane, s1, h1, rw1);
        else if constexpr (SP) c16s_chain<RT, 4, 8, NP>(acc, xp, s1, h1, rw1);
        else c16m_chain<RT, 4, 8>(acc, x, w1, lane, g1);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            acc[t][0] += pq[t][0][0] + pq[t][1][0];
            acc[t][1] += pq[t][0][1] + pq[t][1][1];
        }
        STAMP16(2);
        if constexpr (XL) {
            c16s_prime(h1, s3);
            c16s_publish<RT, true, NP>(acc, xs1, wave, lane);
            STAMP16(3);
#pragma unroll
            for (int t = 0; t < RT; ++t) { acc[t][0] = tb2[0]; acc[t][1] = tb2[1]; }
            c16s_chain_lds<RT, 0, 8, NP>(acc, xs1, lane, s2, h2, rw2);   // layer 2
            STAMP16(4);
            c16s_publish<RT, true, NP>(acc, xs0, wave, lane);        // (xch0: every wave is past its layer-1 reads -- the barrier of the publish before)
#pragma unroll
            for (int t = 0; t < RT; ++t) { acc[t][0] = tb3[0]; acc[t][1] = tb3[1]; }
            c16s_chain_lds<RT, 0, 8, NP>(acc, xs0, lane, s3, h1, rw3);   // layer 3
        } else if constexpr (SP) {                                   // (the ReLUs are folded into the split of the exchange)
            c16s_prime(h1, s3);
            c16s_exchange<RT, true, NP>(xp, acc, xs1, wave, lane);
            STAMP16(3);
#pragma unroll
            for (int t = 0; t < RT; ++t) { acc[t][0] = tb2[0]; acc[t][1] = tb2[1]; }
            c16s_chain<RT, 0, 8, NP>(acc, xp, s2, h2, rw2);          // layer 2
            STAMP16(4);
            c16s_exchange<RT, true, NP>(xp, acc, xs0, wave, lane);
#pragma unroll
            for (int t = 0; t < RT; ++t) { acc[t][0] = tb3[0]; acc[t][1] = tb3[1]; }
            c16s_chain<RT, 0, 8, NP>(acc, xp, s3, h1, rw3);          // layer 3
        } else {
#pragma unroll
        for (int t = 0; t < RT; ++t) c16_relu(acc[t]);
        c16_prime(g1, w3, lane);
        c16m_exchange<RT>(x, acc, xch1, wave, lane);                 // x becomes the exchange result from here on
        STAMP16(3);
#pragma unroll
        for (int t = 0; t < RT; ++t) { acc[t][0] = tb2[0]; acc[t][1] = tb2[1]; }
        c16m_chain<RT>(acc, x, w2, lane, g2);                        // layer 2
        STAMP16(4);
#pragma unroll
        for (int t = 0; t < RT; ++t) c16_relu(acc[t]);
        c16m_exchange<RT>(x, acc, xch0, wave, lane);
#pragma unroll
        for (int t = 0; t < RT; ++t) { acc[t][0] = tb3[0]; acc[t][1] = tb3[1]; }
        c16m_chain<RT>(acc, x, w3, lane, g1);                        // layer 3
        }
        STAMP16(5);
        c16m_layer_norm<RT>(acc, red, tg, tb, wave, n, tabs + T_LN * L);              // acc = this wave's slice of e'
        STAMP16(6);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            if (valid[t]) {                                          // e <- e + e'
                c16_st_tile<BF>(a.Elat, tile[t], row[t], 2 * wave, q, xs[t][0] + acc[t][0]);
                c16_st_tile<BF>(a.Elat, tile[t], row[t], 2 * wave + 1, q, xs[t][1] + acc[t][1]);
            }
            // segmented sum over runs of equal receiver within the 16-edge tile: the 16 rows of a tile are one DPP row (the same in all four lane groups)
            const float m1 = sm1[t], m2 = sm2[t], m4 = sm4[t], m8 = sm8[t];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[t][j][i];
                    v = __builtin_fmaf(dpp_zero<0x111, 0xF>(v), m1, v);
                    v = __builtin_fmaf(dpp_zero<0x112, 0xF>(v), m2, v);
                    v = __builtin_fmaf(dpp_zero<0x114, 0xF>(v), m4, v);
                    v = __builtin_fmaf(dpp_zero<0x118, 0xF>(v), m8, v);
                    acc[t][j][i] = v;
                }
            if (tailf[t]) {
                if (to_carry[t]) {
                    const int64_t cr = (int64_t)2 * ht[t] + (sl_[t] ? 0 : 1);
                    c16_st_row<BF>(a.CARRY, cr, 2 * wave, q, acc[t][0]);
                    c16_st_row<BF>(a.CARRY, cr, 2 * wave + 1, q, acc[t][1]);
                } else {
                    c16_st_tile<BF>(a.AGG, r_[t] >> 5, r_[t] & 31, 2 * wave, q, acc[t][0]);
                    c16_st_tile<BF>(a.AGG, r_[t] >> 5, r_[t] & 31, 2 * wave + 1, q, acc[t][1]);
                }
            }
        }
        STAMP16(7);
        __syncthreads();   // xch0 is rewritten by the next group's first exchange
    }
}

// chunk[0]=W2 [1]=W3 [2]=W1v [3]=W1a [4]=WP [5]=WQ, with two edge sets [6]=W1a2 [7]=WP2 [8]=WQ2.  mode as in NodeArgs.
// Every wave requests only its own two blocks of the v rows and of the aggregate rows (the residual's slice; a launch starts with
// cold caches and a wave cannot keep dozens of cold loads in flight) and the four waves assemble the full rows through LDS; the
// LayerNorm statistics are per-wave partials combined through LDS (c16m_layer_norm).  SETS = 2: the second set's aggregate is one
// more layer-1 chain, and mode 1 projects P, Q of both sets.
// this wave's slice of a node's aggregated messages: the node's AGG slot, or carry rows when its run of edges straddles tiles
// CSH: log2 of the edge kernel's tile height -- 4 with the 16-row edge kernel (carry rows per 16-edge tile), 5 when the edge step ran a
// 32-row kernel (k_edge_ring on a mid-size mesh: carry rows per 32-edge tile)
template <bool BF, int CSH = 4>
DEVINL void c16_agg_slice(f32x4 (&as)[2], const int32_t* rowptr, const float* AGG, const float* CARRY, int64_t zero_row, bool valid, int nn,
                          int64_t tile, int row, int wave, int q) {
    const int a0 = rowptr[nn], a1 = rowptr[nn + 1];
    const int T1 = a0 >> CSH, T2 = (a1 - 1) >> CSH;
    const int extra = (valid && a1 > a0 && T2 > T1) ? (T2 - T1) : 0;
    const bool from_agg = valid && (a1 > a0) && !extra;
    if constexpr (BF) {
        const int64_t cr = extra ? (int64_t)(2 * T1 + 1) : zero_row;
        const uint16_t* A16 = reinterpret_cast<const uint16_t*>(AGG) + c16_bf_tile_off(tile, row, 2 * wave, q);
        const uint16_t* C16 = reinterpret_cast<const uint16_t*>(CARRY) + c16_bf_row_off(cr, 2 * wave, q);
        const uint16_t* src = from_agg ? A16 : C16;
        as[0] = c16_unpack4(*reinterpret_cast<const uint2*>(src));
        as[1] = c16_unpack4(*reinterpret_cast<const uint2*>(src + (from_agg ? 64 * 8 : STRIDE_PROW * 8)));      // next feature block: s + 1
    } else {
        const f32x4* A4 = reinterpret_cast<const f32x4*>(AGG) + c16_tile_idx(tile, row, 2 * wave, q);
        const f32x4* C4 = reinterpret_cast<const f32x4*>(CARRY) + prow_f4(extra ? (int64_t)(2 * T1 + 1) : zero_row, q + 4 * (2 * wave), 128);
        const f32x4* src = from_agg ? A4 : C4;
        as[0] = src[0];
        as[1] = src[from_agg ? 128 : 2 * STRIDE_PROW];                       // the next feature block: four 16-byte pieces = two 32-byte pieces on
    }
    for (int k = 1; __any(k <= extra); ++k)
        if (k <= extra) {
            as[0] += c16_ld_row<BF>(CARRY, (int64_t)2 * (T1 + k), 2 * wave, q);
            as[1] += c16_ld_row<BF>(CARRY, (int64_t)2 * (T1 + k), 2 * wave + 1, q);
        }
}
// split path: one slice -> its pieces in an exchange buffer / all four k-steps' pieces back (no barrier: the caller's)
template <bool RELU, int NP>
DEVINL void c16s_put(u32x4* xch, const f32x4 (&mine)[2], int wave, int lane) {
    const C16X<NP> p = c16s_split<RELU, NP>(mine);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) xch[(wave * NP + pc) * 64 + lane] = p.p[pc];
    if constexpr (NP == 2) c16s_scales<1, NP>(xch)[wave * 16 + (lane & 15)] = p.rs;
}
template <int NP>
DEVINL void c16s_get(C16X<NP> (&full)[1][4], const u32x4* xch, int lane) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        C16X<NP> x[1];
        c16s_fetch<1, NP>(x, xch, ks, lane);
        full[0][ks] = x[0];
    }
}
template <bool BF, int NP>
DEVINL void c16s_project(const C16X<NP> (&v)[1][4], const u32x4* wp, const u32x4* wq, float rwp, float rwq, const float* bq, float* P, float* Q,
                         bool valid, int nn, int wave, int q, C16SRingT<NP>& ga, C16SRingT<NP>& gb, const u32x4* next_wp) {
    f32x4 o[1][2];
    o[0][0] = o[0][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    c16s_prime(gb, wq);
    c16s_chain<1, 0, 8, NP>(o, v, wp, ga, rwp);
    if (valid) {
        c16_st_row<BF>(P, nn, 2 * wave, q, o[0][0]);
        c16_st_row<BF>(P, nn, 2 * wave + 1, q, o[0][1]);
    }
    c16_tab(o[0], bq, wave, q);
    if (next_wp) c16s_prime(ga, next_wp);
    c16s_chain<1, 0, 8, NP>(o, v, wq, gb, rwq);
    if (valid) {
        c16_st_row<BF>(Q, nn, 2 * wave, q, o[0][0]);
        c16_st_row<BF>(Q, nn, 2 * wave + 1, q, o[0][1]);
    }
}
// P = v W_P, Q = v W_Q + bq for this wave's blocks of one edge set (ga primed with wp's first fragments; gb is primed here)
template <bool BF>
DEVINL void c16_project(const f32x4 (&v)[1][8], const float* wp, const float* wq, const float* bq, float* P, float* Q, bool valid, int nn,
                        int wave, int lane, int q, C16Ring& ga, C16Ring& gb, const float* next_wp) {
    f32x4 o[1][2];
    o[0][0] = o[0][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    c16_prime(gb, wq, lane);
    c16m_chain<1>(o, v, wp, lane, ga);
    if (valid) {
        c16_st_row<BF>(P, nn, 2 * wave, q, o[0][0]);
        c16_st_row<BF>(P, nn, 2 * wave + 1, q, o[0][1]);
    }
    c16_tab(o[0], bq, wave, q);
    if (next_wp) c16_prime(ga, next_wp, lane);
    c16m_chain<1>(o, v, wq, lane, gb);
    if (valid) {
        c16_st_row<BF>(Q, nn, 2 * wave, q, o[0][0]);
        c16_st_row<BF>(Q, nn, 2 * wave + 1, q, o[0][1]);
    }
}
template <int SETS, bool BF, int SP = 0, int CSH = 4>      // SP as in k_edge_coop16m
__global__ __launch_bounds__(256, 2) void k_node_coop16(const NodeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NP = SP == 2 ? 2 : 3;
    constexpr int XCH = SP ? 4 * NP * 64 + (SP == 2 ? 16 : 0) : 8 * 64;   // 16-byte slots of an exchange buffer (SP: pieces + 1 / scale per (k-step, row))
    f32x4* xch0 = reinterpret_cast<f32x4*>(smem);
    f32x4* xch1 = xch0 + XCH;
    f32x4* xch2 = xch1 + XCH;
    f32x4* xch3 = xch2 + XCH;                                        // (SETS == 2 only)
    float* red = reinterpret_cast<float*>(xch2 + (SETS == 2 ? 2 : 1) * XCH);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* tabs = a.tabs + T_COUNT * L;
    auto wt = [&](int ch) { return a.chunk[ch] + 2 * C16_CH + wave * 4096; };

    const int ht0 = 2 * a.tile0, nht = 2 * a.ntiles;
    for (int hi = blockIdx.x; hi < nht; hi += gridDim.x) {
        STAMP16(0);
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int n = lane & 15, q = lane >> 4;
        auto ws = [&](int ch) { return c16s_w(SP == 2 ? a.split16h[ch] : a.split16[ch], wave, lane); };    // (SP) the chunk's pieces, this wave's blocks
        auto rw = [&](int ch) { return a.h2_rs[ch]; };                 // (SP = 2) 1 / the chunk's scale
        const int ht = ht0 + hi;
        const int node = ht * 16 + n;
        if (ht * 16 >= a.n) break;
        const bool valid = node < a.n;
        const int nn = valid ? node : a.n - 1;
        const int64_t tile = ht >> 1;
        const int row = 16 * (ht & 1) + n;
        f32x4 v[1][8], full[1][8], acc[1][2], vs[1][2];
        vs[0][0] = c16_ld_tile<BF>(a.V, tile, row, 2 * wave, q);
        vs[0][1] = c16_ld_tile<BF>(a.V, tile, row, 2 * wave + 1, q);
        C16Ring ga, gb;
        if constexpr (SP) {
            C16SRingT<NP> ha, hb;
            C16X<NP> vp[1][4], fp[1][4];
            u32x4* const x0 = reinterpret_cast<u32x4*>(xch0);
            u32x4* const x1 = reinterpret_cast<u32x4*>(xch1);
            u32x4* const x2 = reinterpret_cast<u32x4*>(xch2);
            u32x4* const x3 = reinterpret_cast<u32x4*>(xch3);
            if (a.mode != 2) {
                f32x4 as[2], as2[2];
                c16_agg_slice<BF, CSH>(as, a.rowptr, a.AGG, a.CARRY, a.zero_row, valid, nn, tile, row, wave, q);
                if constexpr (SETS == 2) c16_agg_slice<BF, CSH>(as2, a.rowptr2, a.AGG2, a.CARRY2, a.zero_row2, valid, nn, tile, row, wave, q);
                c16s_prime(ha, ws(2));
                c16_tab(acc[0], tabs + T_B1 * L, wave, q);
                c16s_put<false, NP>(x0, vs[0], wave, lane);
                c16s_put<false, NP>(x1, as, wave, lane);
                if constexpr (SETS == 2) c16s_put<false, NP>(x2, as2, wave, lane);
                __syncthreads();
                c16s_get(vp, x0, lane);
                __builtin_amdgcn_sched_barrier(0);
                STAMP16(1);
                c16s_chain<1, 0, 4, NP>(acc, vp, ws(2), ha, rw(2));      // layer 1, node part
                f32x4 tb2[2], tb3[2], tg[2], tb[2];
                c16_tab(tb2, tabs + T_B2 * L, wave, q);
                c16_tab(tb3, tabs + T_B3 * L, wave, q);
                c16_tab(tg, tabs + T_GAMMA * L, wave, q);
                c16_tab(tb, tabs + T_BETA * L, wave, q);
                c16s_prime(hb, ws(3));
                c16s_chain<1, 4, 8, NP>(acc, vp, ws(2), ha, rw(2));
                STAMP16(2);
                c16s_prime(ha, ws(SETS == 2 ? 6 : 0));
                __builtin_amdgcn_sched_barrier(0);                       // (the aggregate's pieces only now: 48 registers the node part no longer needs)
                c16s_get(fp, x1, lane);
                c16s_chain<1, 0, 8, NP>(acc, fp, ws(3), hb, rw(3));      // layer 1, aggregate part
                if constexpr (SETS == 2) {
                    c16s_get(fp, x2, lane);
                    c16s_prime(hb, ws(0));
                    c16s_chain<1, 0, 8, NP>(acc, fp, ws(6), ha, rw(6));  // layer 1, the second set's aggregate
                }
                C16SRingT<NP>& r2 = SETS == 2 ? hb : ha;                 // the ring that holds layer 2's first fragments
                C16SRingT<NP>& r3 = SETS == 2 ? ha : hb;
                STAMP16(3);
                c16s_prime(r3, ws(1));
                c16s_put<true, NP>(SETS == 2 ? x3 : x2, acc[0], wave, lane); // (ReLU folded into the split)
                __syncthreads();
                c16s_get(fp, SETS == 2 ? x3 : x2, lane);
                acc[0][0] = tb2[0];
                acc[0][1] = tb2[1];
                c16s_chain<1, 0, 8, NP>(acc, fp, ws(0), r2, rw(0));      // layer 2
                STAMP16(4);
                if (a.mode == 1) c16s_prime(r2, ws(4));                  // the projection's first fragments
                c16s_put<true, NP>(x0, acc[0], wave, lane);
                __syncthreads();
                c16s_get(fp, x0, lane);
                acc[0][0] = tb3[0];
                acc[0][1] = tb3[1];
                c16s_chain<1, 0, 8, NP>(acc, fp, ws(1), r3, rw(1));      // layer 3
                c16m_layer_norm<1>(acc, red, tg, tb, wave, n, tabs + T_LN * L);
                STAMP16(5);
                acc[0][0] = c16_round<BF>(acc[0][0] + vs[0][0]);         // v <- v + v'
                acc[0][1] = c16_round<BF>(acc[0][1] + vs[0][1]);
                if (valid) {
                    c16_st_tile<BF>(a.V, tile, row, 2 * wave, q, acc[0][0]);
                    c16_st_tile<BF>(a.V, tile, row, 2 * wave + 1, q, acc[0][1]);
                }
                if (a.mode == 1) {                                       // P, Q of the next step, on the updated rows
                    c16s_put<false, NP>(x1, acc[0], wave, lane);
                    __syncthreads();
                    c16s_get(vp, x1, lane);
                    STAMP16(6);
                    c16s_project<BF, NP>(vp, ws(4), ws(5), rw(4), rw(5), tabs + T_BQ * L, a.P, a.Q, valid, nn, wave, q, r2, r3, SETS == 2 ? ws(7) : nullptr);
                    if constexpr (SETS == 2)
                        c16s_project<BF, NP>(vp, ws(7), ws(8), rw(7), rw(8), a.tabs2 + (T_COUNT + T_BQ) * L, a.P2, a.Q2, valid, nn, wave, q, r2, r3, nullptr);
                }
            } else {                                                     // projection only (before the first step; one set per launch)
                c16s_prime(ha, ws(4));
                c16s_put<false, NP>(x0, vs[0], wave, lane);
                __syncthreads();
                c16s_get(vp, x0, lane);
                c16s_project<BF, NP>(vp, ws(4), ws(5), rw(4), rw(5), tabs + T_BQ * L, a.P, a.Q, valid, nn, wave, q, ha, hb, nullptr);
            }
        } else if (a.mode != 2) {
            f32x4 as[2], as2[2];
            c16_agg_slice<BF, CSH>(as, a.rowptr, a.AGG, a.CARRY, a.zero_row, valid, nn, tile, row, wave, q);
            if constexpr (SETS == 2) c16_agg_slice<BF, CSH>(as2, a.rowptr2, a.AGG2, a.CARRY2, a.zero_row2, valid, nn, tile, row, wave, q);
            c16_prime(ga, wt(2), lane);
            c16_tab(acc[0], tabs + T_B1 * L, wave, q);
            xch0[(2 * wave) * 64 + lane] = vs[0][0];
            xch0[(2 * wave + 1) * 64 + lane] = vs[0][1];
            xch1[(2 * wave) * 64 + lane] = as[0];
            xch1[(2 * wave + 1) * 64 + lane] = as[1];
            if constexpr (SETS == 2) {
                xch2[(2 * wave) * 64 + lane] = as2[0];
                xch2[(2 * wave + 1) * 64 + lane] = as2[1];
            }
            __syncthreads();
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) v[0][bb] = xch0[bb * 64 + lane];
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) full[0][bb] = xch1[bb * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            STAMP16(1);
            c16m_chain<1, 0, 4>(acc, v, wt(2), lane, ga);            // layer 1, node part
            f32x4 tb2[2], tb3[2], tg[2], tb[2];
            c16_tab(tb2, tabs + T_B2 * L, wave, q);
            c16_tab(tb3, tabs + T_B3 * L, wave, q);
            c16_tab(tg, tabs + T_GAMMA * L, wave, q);
            c16_tab(tb, tabs + T_BETA * L, wave, q);
            c16_prime(gb, wt(3), lane);
            c16m_chain<1, 4, 8>(acc, v, wt(2), lane, ga);
            STAMP16(2);
            c16_prime(ga, wt(SETS == 2 ? 6 : 0), lane);
            c16m_chain<1>(acc, full, wt(3), lane, gb);               // layer 1, aggregate part
            if constexpr (SETS == 2) {
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) full[0][bb] = xch2[bb * 64 + lane];
                c16_prime(gb, wt(0), lane);
                c16m_chain<1>(acc, full, wt(6), lane, ga);           // layer 1, the second set's aggregate
            }
            C16Ring& r2 = SETS == 2 ? gb : ga;                       // the ring that holds layer 2's first fragments
            C16Ring& r3 = SETS == 2 ? ga : gb;
            STAMP16(3);
            c16_relu(acc[0]);
            c16_prime(r3, wt(1), lane);
            c16m_exchange<1>(full, acc, SETS == 2 ? xch3 : xch2, wave, lane);
            acc[0][0] = tb2[0];
            acc[0][1] = tb2[1];
            c16m_chain<1>(acc, full, wt(0), lane, r2);               // layer 2
            STAMP16(4);
            c16_relu(acc[0]);
            if (a.mode == 1) c16_prime(r2, wt(4), lane);             // the projection's first fragments
            c16m_exchange<1>(full, acc, xch0, wave, lane);
            acc[0][0] = tb3[0];
            acc[0][1] = tb3[1];
            c16m_chain<1>(acc, full, wt(1), lane, r3);               // layer 3
            c16m_layer_norm<1>(acc, red, tg, tb, wave, n, tabs + T_LN * L);
            STAMP16(5);
            acc[0][0] = c16_round<BF>(acc[0][0] + vs[0][0]);         // v <- v + v'  (this wave's slice; the value the store keeps)
            acc[0][1] = c16_round<BF>(acc[0][1] + vs[0][1]);
            if (valid) {
                c16_st_tile<BF>(a.V, tile, row, 2 * wave, q, acc[0][0]);
                c16_st_tile<BF>(a.V, tile, row, 2 * wave + 1, q, acc[0][1]);
            }
            if (a.mode == 1) {                                       // P, Q of the next step, on the updated rows
                c16m_exchange<1>(v, acc, xch1, wave, lane);
                STAMP16(6);
                c16_project<BF>(v, wt(4), wt(5), tabs + T_BQ * L, a.P, a.Q, valid, nn, wave, lane, q, r2, r3, SETS == 2 ? wt(7) : nullptr);
                if constexpr (SETS == 2)
                    c16_project<BF>(v, wt(7), wt(8), a.tabs2 + (T_COUNT + T_BQ) * L, a.P2, a.Q2, valid, nn, wave, lane, q, r2, r3, nullptr);
            }
        } else {                                                     // projection only (before the first step; one set per launch)
            c16_prime(ga, wt(4), lane);
            c16m_exchange<1>(v, vs, xch0, wave, lane);
            c16_project<BF>(v, wt(4), wt(5), tabs + T_BQ * L, a.P, a.Q, valid, nn, wave, lane, q, ga, gb, nullptr);
        }
        STAMP16(7);
        __syncthreads();
    }
}

// first dense layer with a tiny input width: acc[feature] = b1 + sum_k x[k] * W1[k][feature]  (VALU)
template <int NT>
DEVINL void first_layer(f32x16 (&acc)[NT], const float* w1f, int k, float xk, int h) {
    constexpr int L = 32 * NT;
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w1f + (int64_t)k * L) + h;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w = w4[2 * (4 * t + g)];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = fmaf(xk, w[i], acc[t][4 * g + i]);
        }
}

// One chunk of an encoder / decoder chain: acc += in W on the fp32 MFMA pipe (fragment-order chunk in LDS or L2) or, H2 (round 5: L = 128,
// the split path's switch), on the chunk's two fp16 pieces (split_common.hpp: h2_chunk_inplace; same 64 KiB per chunk, LDS or L2).
template <int NT, bool RES, bool H2>
DEVINL void enc_chunk(f32x16 (&acc)[NT], const f32x16 (&in)[NT], const float* w, int lane, float rsw) {
    if constexpr (H2) {
        static_assert(NT == 4, "the fp16 pieces exist at L = 128");
        h2_chunk_inplace(acc, in, w, lane, rsw);
    } else {
        mfma_chunk<NT, RES>(acc, in, w, lane);
    }
}
// the cooperative kernels' chains: this wave's feature block of acc += in W (fp32: t-major copy through CoopRing; H2: the pieces through H2CoopRing)
template <bool H2> struct ERing;
template <> struct ERing<false> { CoopRing r; };
template <> struct ERing<true> { H2CoopRing r; };
template <bool H2>
DEVINL void e_prime(ERing<H2>& g, const float* chunk, const uint16_t* pieces, int tq, int lane) {
    if constexpr (H2) h2c_prime(g.r, reinterpret_cast<const u32x4*>(pieces) + tq * 64 + lane);
    else coop_prime(g.r, chunk + 128 * 128 + tq * 4096, lane);
}
template <bool H2, bool FENCE>
DEVINL void e_chain_primed(f32x16& acc, const f32x16 (&in)[4], const float* chunk, const uint16_t* pieces, float rsw, int tq, int lane, ERing<H2>& g) {
    if constexpr (H2) h2c_chain_primed(acc, in, reinterpret_cast<const u32x4*>(pieces) + tq * 64 + lane, g.r, rsw);
    else coop_chain_primed<FENCE>(acc, in, chunk + 128 * 128 + tq * 4096, lane, g.r);
}

// ================================================================================================
// Encoder, node side (K0a+K1) + projection of step-0 P,Q.  chunk[0]=W2 [1]=W3 [2]=WP [3]=WQ
// ================================================================================================
template <int NT, int NRES, bool GEN = false, bool H2 = false>
__global__ __launch_bounds__(512, 2) void k_enc_node(const EncNodeArgs a) {
    constexpr int L = 32 * NT, CH = 16 * NT * 64 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
#pragma unroll
    for (int r = 0; r < NRES; ++r) copy_to_lds(smem + r * CH, H2 ? reinterpret_cast<const float*>(a.splith[r]) : a.chunk[r], CH);
    float* tb = smem + NRES * CH;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* w2 = NRES > 0 ? smem : (H2 ? reinterpret_cast<const float*>(a.splith[0]) : a.chunk[0]);
    const float* w3 = NRES > 1 ? smem + CH : (H2 ? reinterpret_cast<const float*>(a.splith[1]) : a.chunk[1]);
    const float* wp = NRES > 2 ? smem + 2 * CH : (H2 ? reinterpret_cast<const float*>(a.splith[2]) : a.chunk[2]);
    const float* wq = NRES > 3 ? smem + 3 * CH : (H2 ? reinterpret_cast<const float*>(a.splith[3]) : a.chunk[3]);

    for (TileWalk tw(a.ntiles, wave); tw.tile < tw.end; tw.tile += tw.stride) {
        OPAQUE_LANE();
        const int n = tw.tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        const int64_t g = a.gid ? a.gid[nn] : nn;     // null gid: the source rows are already in local order
        f32x16 acc[NT], y[NT];
        tab_frag<NT>(acc, tb + T_B1 * L, h);
        const int Fn = a.wa + a.wb;
        for (int k = 0; k < Fn; ++k) {
            float xk = (k < a.wa) ? a.srcA[g * a.wa + k] : a.srcB[g * a.wb + (k - a.wa)];
            if (a.scale) xk = fmaf(xk, a.scale[k], a.shift[k]);
            first_layer<NT>(acc, a.w1f, k, xk, h);
        }
        if constexpr (GEN) {
            gen_hidden<NT>(acc, y, a.gen, lane, h);
            gen_final<NT>(acc, y, a.gen, lane, h);
        } else {
            relu_frag<NT>(acc);
            tab_frag<NT>(y, tb + T_B2 * L, h);
            enc_chunk<NT, (NRES > 0), H2>(y, acc, w2, lane, a.h2_rs[0]);
            relu_frag<NT>(y);
            tab_frag<NT>(acc, tb + T_B3 * L, h);
            enc_chunk<NT, (NRES > 1), H2>(acc, y, w3, lane, a.h2_rs[1]);
        }
        layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);
        if (!valid) zero_frag<NT>(acc);                          // padding rows stay zero (checksums)
        store_frag<NT>(tile_ptr(a.V, tw.tile, L, lane), STRIDE_TILE, acc);
        zero_frag<NT>(y);
        enc_chunk<NT, (NRES > 2), H2>(y, acc, wp, lane, a.h2_rs[2]);
        if (valid) store_frag<NT>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, y);
        tab_frag<NT>(y, tb + T_BQ * L, h);
        enc_chunk<NT, (NRES > 3), H2>(y, acc, wq, lane, a.h2_rs[3]);
        if (valid) store_frag<NT>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, y);
    }
}

// ================================================================================================
// Cooperative node encoder and decoder for small meshes (L = 128): 4 waves per tile like k_node_coop.  On a cylinder-sized
// mesh the one-wave-per-tile encoder / decoder cost 60 + 32 us of every 840 us right-hand side of a rollout.
// ================================================================================================
// chunk[0]=W2 [1]=W3 [2]=WP [3]=WQ; the t-major copy of a chunk follows its fragment-major copy (mgn_set_params)
template <bool FENCE, bool H2 = false>
__global__ __launch_bounds__(256, 2) void k_enc_node_coop(const EncNodeArgs a) {
    constexpr int L = 128, CH = L * L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xch0 = reinterpret_cast<f32x4*>(smem);
    f32x4* xch1 = xch0 + 16 * 64;
    float* tb = smem + 2 * 16 * 64 * 4;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tq = wave;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        OPAQUE_LANE();
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        const int64_t g = a.gid ? a.gid[nn] : nn;     // null gid: the source rows are already in local order
        f32x16 in[4], acc;
        ERing<H2> r2, r3;
        e_prime<H2>(r2, a.chunk[0], a.splith[0], tq, lane);
        // layer 1 (a handful of input features: VALU), every wave the full row
        tab_frag<4>(in, tb + T_B1 * L, h);
        const int Fn = a.wa + a.wb;
        for (int k = 0; k < Fn; ++k) {
            float xk = (k < a.wa) ? a.srcA[g * a.wa + k] : a.srcB[g * a.wb + (k - a.wa)];
            if (a.scale) xk = fmaf(xk, a.scale[k], a.shift[k]);
            first_layer<4>(in, a.w1f, k, xk, h);
        }
        relu_frag<4>(in);
        tab_quarter(acc, tb + T_B2 * L, tq, h);
        e_prime<H2>(r3, a.chunk[1], a.splith[1], tq, lane);
        e_chain_primed<H2, FENCE>(acc, in, a.chunk[0], a.splith[0], a.h2_rs[0], tq, lane, r2);                            // layer 2
        relu_quarter(acc);
        coop_exchange(in, acc, xch0, wave, lane);
        tab_quarter(acc, tb + T_B3 * L, tq, h);
        e_prime<H2>(r2, a.chunk[2], a.splith[2], tq, lane);
        e_chain_primed<H2, FENCE>(acc, in, a.chunk[1], a.splith[1], a.h2_rs[1], tq, lane, r3);                            // layer 3
        coop_exchange(in, acc, xch1, wave, lane);
        coop_layer_norm(acc, in, tb + T_GAMMA * L, tb + T_BETA * L, tq, h);
        if (!valid) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;                        // padding rows stay zero (checksums)
        }
        store_quarter(tile_ptr(a.V, tile, L, lane), STRIDE_TILE, tq, acc);
        e_prime<H2>(r3, a.chunk[3], a.splith[3], tq, lane);
        coop_exchange(in, acc, xch0, wave, lane);                            // full latent row for the projection
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        e_chain_primed<H2, FENCE>(acc, in, a.chunk[2], a.splith[2], a.h2_rs[2], tq, lane, r2);
        if (valid) store_quarter(prow_ptr(a.P, nn, L, h), STRIDE_PROW, tq, acc);
        tab_quarter(acc, tb + T_BQ * L, tq, h);
        e_chain_primed<H2, FENCE>(acc, in, a.chunk[3], a.splith[3], a.h2_rs[3], tq, lane, r3);
        if (valid) store_quarter(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, tq, acc);
        __syncthreads();
    }
}

// edge encoder: chunk[0]=W2 [1]=W3
template <bool FENCE, bool H2 = false>
__global__ __launch_bounds__(256, 2) void k_enc_edge_coop(const EncEdgeArgs a) {
    constexpr int L = 128, CH = L * L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xch0 = reinterpret_cast<f32x4*>(smem);
    f32x4* xch1 = xch0 + 16 * 64;
    float* tb = smem + 2 * 16 * 64 * 4;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tq = wave;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        OPAQUE_LANE();
        const int64_t eid = (int64_t)tile * TILE + c;
        const bool valid = eid < a.E;
        const int64_t g = a.gid ? a.gid[valid ? eid : 0] : (valid ? eid : 0);
        f32x16 in[4], acc;
        ERing<H2> r2, r3;
        e_prime<H2>(r2, a.chunk[0], a.splith[0], tq, lane);
        tab_frag<4>(in, tb + T_B1 * L, h);
        for (int k = 0; k < a.Fe; ++k) {
            float xk = a.ef[g * a.Fe + k];
            if (a.scale) xk = fmaf(xk, a.scale[k], a.shift[k]);
            first_layer<4>(in, a.w1f, k, xk, h);
        }
        relu_frag<4>(in);
        tab_quarter(acc, tb + T_B2 * L, tq, h);
        e_prime<H2>(r3, a.chunk[1], a.splith[1], tq, lane);
        e_chain_primed<H2, FENCE>(acc, in, a.chunk[0], a.splith[0], a.h2_rs[0], tq, lane, r2);
        relu_quarter(acc);
        coop_exchange(in, acc, xch0, wave, lane);
        tab_quarter(acc, tb + T_B3 * L, tq, h);
        e_chain_primed<H2, FENCE>(acc, in, a.chunk[1], a.splith[1], a.h2_rs[1], tq, lane, r3);
        coop_exchange(in, acc, xch1, wave, lane);
        coop_layer_norm(acc, in, tb + T_GAMMA * L, tb + T_BETA * L, tq, h);
        if (!valid) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        }
        store_quarter(tile_ptr(a.Elat, tile, L, lane), STRIDE_TILE, tq, acc);
        __syncthreads();
    }
}

// chunk[0]=W1 [1]=W2; last layer (L -> O) by wave 0 from the exchanged row
template <bool FENCE, bool H2 = false>
__global__ __launch_bounds__(256, 2) void k_decode_coop(const DecArgs a) {
    constexpr int L = 128, CH = L * L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xch0 = reinterpret_cast<f32x4*>(smem);
    f32x4* xch1 = xch0 + 16 * 64;
    float* tb = smem + 2 * 16 * 64 * 4;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tq = wave;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        OPAQUE_LANE();
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        f32x16 in[4], acc;
        ERing<H2> r2;
        e_prime<H2>(r2, a.chunk[1], a.splith[1], tq, lane);
        load_frag<4>(in, tile_ptr(a.V, tile, L, lane), STRIDE_TILE);
        tab_quarter(acc, tb + T_B1 * L, tq, h);
        { ERing<H2> r1; e_prime<H2>(r1, a.chunk[0], a.splith[0], tq, lane); __builtin_amdgcn_sched_barrier(0); e_chain_primed<H2, FENCE>(acc, in, a.chunk[0], a.splith[0], a.h2_rs[0], tq, lane, r1); }
        relu_quarter(acc);
        coop_exchange(in, acc, xch0, wave, lane);
        tab_quarter(acc, tb + T_B2 * L, tq, h);
        e_chain_primed<H2, FENCE>(acc, in, a.chunk[1], a.splith[1], a.h2_rs[1], tq, lane, r2);
        relu_quarter(acc);
        coop_exchange(in, acc, xch1, wave, lane);
        if (wave == 0) {
            const float m = a.mask ? a.mask[a.gid ? a.gid[nn] : nn] : 1.0f;
            for (int o = 0; o < a.O; ++o) {
                const f32x4* w4 = reinterpret_cast<const f32x4*>(a.w3f + (int64_t)o * L) + h;
                float sacc = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 w = w4[2 * (4 * t + g)];
#pragma unroll
                        for (int i = 0; i < 4; ++i) sacc = fmaf(in[t][4 * g + i], w[i], sacc);
                    }
                sacc += __shfl_xor(sacc, 32, 64);
                sacc += a.b3[o];
                if (a.oscale) sacc = fmaf(sacc, a.oscale[o], a.oshift[o]);
                sacc *= m;
                if (valid && h == 0) a.out[(int64_t)nn * a.O + o] = sacc;
            }
        }
        __syncthreads();
    }
}

// ================================================================================================
// Encoder, edge side (K0b+K2).  chunk[0]=W2 [1]=W3
// ================================================================================================
template <int NT, int NRES, bool GEN = false, bool H2 = false>
__global__ __launch_bounds__(512, 2) void k_enc_edge(const EncEdgeArgs a) {
    constexpr int L = 32 * NT, CH = 16 * NT * 64 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
#pragma unroll
    for (int r = 0; r < NRES; ++r) copy_to_lds(smem + r * CH, H2 ? reinterpret_cast<const float*>(a.splith[r]) : a.chunk[r], CH);
    float* tb = smem + NRES * CH;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* w2 = NRES > 0 ? smem : (H2 ? reinterpret_cast<const float*>(a.splith[0]) : a.chunk[0]);
    const float* w3 = NRES > 1 ? smem + CH : (H2 ? reinterpret_cast<const float*>(a.splith[1]) : a.chunk[1]);

    for (TileWalk tw(a.ntiles, wave); tw.tile < tw.end; tw.tile += tw.stride) {
        OPAQUE_LANE();
        const int64_t eid = (int64_t)tw.tile * TILE + c;
        const bool valid = eid < a.E;
        const int64_t ee = valid ? eid : 0;
        const int64_t g = a.gid ? a.gid[ee] : ee;
        f32x16 acc[NT], y[NT];
        tab_frag<NT>(acc, tb + T_B1 * L, h);
        for (int k = 0; k < a.Fe; ++k) {
            float xk = a.ef[g * a.Fe + k];
            if (a.scale) xk = fmaf(xk, a.scale[k], a.shift[k]);
            first_layer<NT>(acc, a.w1f, k, xk, h);
        }
        if constexpr (GEN) {
            gen_hidden<NT>(acc, y, a.gen, lane, h);
            gen_final<NT>(acc, y, a.gen, lane, h);
        } else {
            relu_frag<NT>(acc);
            tab_frag<NT>(y, tb + T_B2 * L, h);
            enc_chunk<NT, (NRES > 0), H2>(y, acc, w2, lane, a.h2_rs[0]);
            relu_frag<NT>(y);
            tab_frag<NT>(acc, tb + T_B3 * L, h);
            enc_chunk<NT, (NRES > 1), H2>(acc, y, w3, lane, a.h2_rs[1]);
        }
        layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);
        if (!valid) zero_frag<NT>(acc);
        store_frag<NT>(tile_ptr(a.Elat, tw.tile, L, lane), STRIDE_TILE, acc);
    }
}

// ================================================================================================
// Decoder (K7) + inverse normaliser + val_mask epilogue (K8).  chunk[0]=W1 [1]=W2
// ================================================================================================
template <int NT, int NRES, bool GEN = false, bool H2 = false>
__global__ __launch_bounds__(512, 2) void k_decode(const DecArgs a) {
    constexpr int L = 32 * NT, CH = 16 * NT * 64 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
#pragma unroll
    for (int r = 0; r < NRES; ++r) copy_to_lds(smem + r * CH, H2 ? reinterpret_cast<const float*>(a.splith[r]) : a.chunk[r], CH);
    float* tb = smem + NRES * CH;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* w1 = NRES > 0 ? smem : (H2 ? reinterpret_cast<const float*>(a.splith[0]) : a.chunk[0]);
    const float* w2 = NRES > 1 ? smem + CH : (H2 ? reinterpret_cast<const float*>(a.splith[1]) : a.chunk[1]);

    for (TileWalk tw(a.ntiles, wave); tw.tile < tw.end; tw.tile += tw.stride) {
        OPAQUE_LANE();
        const int n = tw.tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        f32x16 v[NT], acc[NT];
        load_frag<NT>(v, tile_ptr(a.V, tw.tile, L, lane), STRIDE_TILE);
        tab_frag<NT>(acc, tb + T_B1 * L, h);
        enc_chunk<NT, (NRES > 0), H2>(acc, v, w1, lane, a.h2_rs[0]);
        if constexpr (GEN) {                                    // middle layers, then the L -> O layer below reads v
            gen_hidden<NT>(acc, v, a.gen, lane, h);
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = acc[t];
        } else {
            relu_frag<NT>(acc);
            tab_frag<NT>(v, tb + T_B2 * L, h);
            enc_chunk<NT, (NRES > 1), H2>(v, acc, w2, lane, a.h2_rs[1]);
            relu_frag<NT>(v);
        }
        const float m = a.mask ? a.mask[a.gid ? a.gid[nn] : nn] : 1.0f;
        for (int o = 0; o < a.O; ++o) {
            const f32x4* w4 = reinterpret_cast<const f32x4*>(a.w3f + (int64_t)o * L) + h;
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = w4[2 * (4 * t + g)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) s = fmaf(v[t][4 * g + i], w[i], s);
                }
            s += __shfl_xor(s, 32, 64);
            s += a.b3[o];
            if (a.oscale) s = fmaf(s, a.oscale[o], a.oshift[o]);
            s *= m;
            if (valid && h == 0) a.out[(int64_t)nn * a.O + o] = s;
        }
    }
}

// ================================================================================================
// bf16 processor kernels (L = 128): bf16 storage, v_mfma_f32_32x32x16_bf16, fp32 accumulate / LayerNorm / residual /
// aggregation.  Same lane-per-row design: lane (c,h) owns row c; a k-step s covers 16 features and each half supplies
// 8 of them, element j of half h being feature  f(s,h,j) = 32(s>>1) + 16(s&1) + 8(j>>2) + 4h + (j&3)  -- exactly the
// features this lane's accumulator registers 8(s&1)..8(s&1)+7 of block t = s>>1 hold, so an accumulator becomes the
// next layer's B operand by eight v_cvt_pk_bf16_f32 per k-step and no data movement.  A row is stored as 16 pieces
// of 8 bf16 in that order (piece 2s+h); weights are 32 KiB per chunk, so every chunk of a kernel is LDS-resident.
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BF_STRIDE_ROW = STRIDE_PROW, BF_STRIDE_TILE = 64;   // in 16-byte pieces (rows: blocks of eight, bf_prow_piece)

DEVINL const bf16x8* bf_row_ptr(const uint16_t* base, int64_t row, int h) { return reinterpret_cast<const bf16x8*>(base) + bf_prow_piece(row, h); }
DEVINL bf16x8* bf_row_ptr(uint16_t* base, int64_t row, int h) { return reinterpret_cast<bf16x8*>(base) + bf_prow_piece(row, h); }
DEVINL const bf16x8* bf_tile_ptr(const uint16_t* base, int64_t tile, int lane) { return reinterpret_cast<const bf16x8*>(base + tile * (TILE * 128)) + lane; }
DEVINL bf16x8* bf_tile_ptr(uint16_t* base, int64_t tile, int lane) { return reinterpret_cast<bf16x8*>(base + tile * (TILE * 128)) + lane; }

DEVINL void bf_load(bf16x8 (&x)[8], const bf16x8* __restrict__ p, int stride) {
#pragma unroll
    for (int s = 0; s < 8; ++s) x[s] = p[s * stride];
}
DEVINL void bf_store(bf16x8* __restrict__ p, int stride, const bf16x8 (&x)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) p[s * stride] = x[s];
}
DEVINL void bf_pack(bf16x8 (&x)[8], const f32x16 (&acc)[4]) {          // accumulator -> B operand / storage
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[s][j] = (__bf16)acc[s >> 1][8 * (s & 1) + j];
}
DEVINL void bf_unpack_add(f32x16 (&acc)[4], const bf16x8 (&x)[8]) {    // acc += float(x)
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[s >> 1][8 * (s & 1) + j] += (float)x[s][j];
}

// one 128 x 128 chunk from LDS: w[(s*4 + t)*64 + lane] is the A fragment of k-step s, feature block t
DEVINL void bf_chunk(f32x16 (&acc)[4], const bf16x8 (&in)[8], const bf16x8* w, int lane) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(s * 4 + t) * 64 + lane], in[s], acc[t], 0, 0, 0);
}

constexpr int BF_CH = 128 * 128;   // bf16 elements per chunk (32 KiB)


#ifndef MGN_BF_WAVES
#define MGN_BF_WAVES 12   // waves per block of the bf16 edge kernel: 3 per SIMD (<= 168 VGPRs)
#endif
__global__ __launch_bounds__(MGN_BF_WAVES * 64, MGN_BF_WAVES / 4) void k_edge_bf16(const BfEdgeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
#pragma unroll
    for (int r = 0; r < 3; ++r) copy_to_lds16(wl + r * BF_CH, a.chunk[r], BF_CH, a.ntiles <= 16 * 1024);
    float* tb = smem + 3 * BF_CH / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const bf16x8* w2 = reinterpret_cast<const bf16x8*>(wl);
    const bf16x8* w3 = reinterpret_cast<const bf16x8*>(wl + BF_CH);
    const bf16x8* w1 = reinterpret_cast<const bf16x8*>(wl + 2 * BF_CH);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    EdgeArgs ia{};   // load_edge_idx only needs the index arrays
    ia.snd = a.snd; ia.rcv = a.rcv; ia.E = a.E;
    for (TileWalk tw(a.ntiles, wave); tw.tile < tw.end; tw.tile += tw.stride) {
        OPAQUE_LANE();
        const int tile = a.tile0 + tw.tile;
        const EdgeIdx ix = load_edge_idx(ia, tile, c);
        const bool valid = ix.r >= 0;
        const int r = valid ? ix.r : 0;
        bf16x8 x[8], in[8];
        f32x16 acc[4];   // ONE fp32 accumulator array: the packed copy `in` is the next layer's operand
        bf16x8* etile = bf_tile_ptr(a.Elat, tile, lane);
        bf_load(x, etile, BF_STRIDE_TILE);
        zero_frag<4>(acc);
        bf_load(in, bf_row_ptr(a.P, ix.s, h), BF_STRIDE_ROW);
        bf_unpack_add(acc, in);
        bf_load(in, bf_row_ptr(a.Q, r, h), BF_STRIDE_ROW);
        bf_unpack_add(acc, in);
        bf_chunk(acc, x, w1, lane);                              // layer 1 (edge part; P, Q, b1 preloaded)
        relu_frag<4>(acc);
        bf_pack(in, acc);
        tab_frag<4>(acc, tb + T_B2 * L, h);
        bf_chunk(acc, in, w2, lane);                             // layer 2
        relu_frag<4>(acc);
        bf_pack(in, acc);
        tab_frag<4>(acc, tb + T_B3 * L, h);
        bf_chunk(acc, in, w3, lane);                             // layer 3
        layer_norm_frag<4>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);   // acc = e' (fp32)
        // residual in fp32, stored as bf16
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) in[s][j] = (__bf16)((float)x[s][j] + acc[s >> 1][8 * (s & 1) + j]);
        if (valid) bf_store(etile, BF_STRIDE_TILE, in);
        // segmented sum of e' (fp32) over runs of equal receiver
        const int reff = valid ? r : (-4 - c);
        const int rprev = __shfl_up(reff, 1, 32);
        const int rnext = __shfl_down(reff, 1, 32);
        const bool head = (c == 0) || (reff != rprev);
        const unsigned hm = (unsigned)__ballot(head);
        const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
        const int st_in = max(start, c & 16);
        const bool c1 = (c - 1 >= st_in), c2 = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
        const bool cx = (c >= 16) && (start <= 15);
        segmented_scan<4, true>(acc, c1, c2, c4, c8, cx);
        const bool tail = valid && ((c == 31) || (reff != rnext));
        const int r_first = __builtin_amdgcn_readfirstlane(reff);
        const bool sl = (start == 0) && (ix.r_before == r_first);
        const bool sr = (c == 31) && (ix.r_after == reff);
        const bool to_carry = sl || sr;
        bf16x8* dst = to_carry ? bf_row_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), h)
                               : bf_tile_ptr(a.AGG, r >> 5, 32 * h + (r & 31));
        bf_pack(in, acc);
        if (tail) bf_store(dst, to_carry ? BF_STRIDE_ROW : BF_STRIDE_TILE, in);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// bf16 processor, second generation (round 2).  What the rocprofv3 passes of the first version showed (profiles/r02/
// pmc_summary_bench_1m_bf16.json): 1 500 VALU instructions per 32-edge tile against 96 MFMAs, waves waiting on memory for 68 % of
// their cycles (index -> gathered P / Q rows is a serial chain per tile, nothing is requested ahead), 5.2 GB moved per launch
// against 3.9 GB algorithmic.  Changes:
//   * software pipeline: the operands of tile t+1 (e tile, P[s], Q[r]) are requested at the top of tile t and the indices of
//     tile t+2 with them, so a wave never waits for an index -> gather chain (two waves per SIMD, 256 registers each);
//   * unpack-and-accumulate in ONE instruction: v_dot2c_f32_bf16 with a (1, 0) / (0, 1) selector is float(x.lo / x.hi) + c
//     (exact product, sum within 1 fp32 ulp of IEEE: tools/dot2_probe.hip) -- replaces shift / mask + add for P + Q and the residual;
//   * ReLU on the PACKED row: max(x, 0) of a bf16 is a signed 16-bit integer max with 0 (v_pk_max_i16: one instruction per pair,
//     after the conversion; commutes with the rounding, -0 -> +0);
//   * the edge latents, touched once per step, are read and written non-temporally (they do not displace the gathered P / Q
//     rows from L2).
// ------------------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct BfSel { bf16x2 lo, hi; };
// the selectors go through opaque registers: as compile-time constants hipcc (ROCm 7.2) folds the pair (1.0, 0.0) into the inline
// constant `1.0`, which the instruction reads as the 32-bit pattern 0x3F800000 = the pair (0.0, 1.0) (tools/dot2_probe.hip)
DEVINL BfSel bf_selectors() {
    unsigned u0 = 0x00003F80u, u1 = 0x3F800000u;
    asm volatile("" : "+s"(u0), "+s"(u1));
    BfSel s;
    s.lo = __builtin_bit_cast(bf16x2, u0);
    s.hi = __builtin_bit_cast(bf16x2, u1);
    return s;
}
DEVINL const u32x4* bfq_row_ptr(const uint16_t* base, int64_t row, int h) { return reinterpret_cast<const u32x4*>(base) + bf_prow_piece(row, h); }
DEVINL u32x4* bfq_row_ptr(uint16_t* base, int64_t row, int h) { return reinterpret_cast<u32x4*>(base) + bf_prow_piece(row, h); }
DEVINL const u32x4* bfq_tile_ptr(const uint16_t* base, int64_t tile, int lane) { return reinterpret_cast<const u32x4*>(base + tile * (TILE * 128)) + lane; }
DEVINL u32x4* bfq_tile_ptr(uint16_t* base, int64_t tile, int lane) { return reinterpret_cast<u32x4*>(base + tile * (TILE * 128)) + lane; }

template <bool NT>
DEVINL void bfq_load(u32x4 (&x)[8], const u32x4* __restrict__ p, int stride) {
#pragma unroll
    for (int s = 0; s < 8; ++s) x[s] = NT ? __builtin_nontemporal_load(p + s * stride) : p[s * stride];
}
template <bool NT>
DEVINL void bfq_store(u32x4* __restrict__ p, int stride, const u32x4 (&x)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (NT) __builtin_nontemporal_store(x[s], p + s * stride);
        else p[s * stride] = x[s];
    }
}
// acc (+)= float(x): element j of piece s is register 8(s&1) + j of block s>>1
template <bool INIT>
DEVINL void bfq_acc_add(f32x16 (&acc)[4], const u32x4 (&x)[8], const BfSel& sel) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // through a scalar: clang's __builtin_bit_cast of a vector ELEMENT lvalue copies from the vector's base address, i.e.
            // always element 0 (every pair came out as a copy of the first -- the "80 % off" of round 1's packed-ReLU attempts)
            const unsigned w = x[s][d];
            const bf16x2 v = __builtin_bit_cast(bf16x2, w);
            const int k = 8 * (s & 1) + 2 * d;
            acc[s >> 1][k] = __builtin_amdgcn_fdot2_f32_bf16(v, sel.lo, INIT ? 0.f : acc[s >> 1][k], false);
            acc[s >> 1][k + 1] = __builtin_amdgcn_fdot2_f32_bf16(v, sel.hi, INIT ? 0.f : acc[s >> 1][k + 1], false);
        }
}
DEVINL unsigned bf_pk2(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
// accumulator -> packed row (next B operand / storage); RELU: max(., 0) on the packed pairs
template <bool RELU>
DEVINL void bfq_pack(u32x4 (&x)[8], const f32x16 (&acc)[4]) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned w = bf_pk2(acc[s >> 1][8 * (s & 1) + 2 * d], acc[s >> 1][8 * (s & 1) + 2 * d + 1]);
            if (RELU) asm("v_pk_max_i16 %0, %1, 0" : "=v"(w) : "v"(w));   // reads a VALU result (the conversion), never an MFMA one
            x[s][d] = w;
        }
}
// One 128 x 128 chunk from LDS.  Every MFMA (32 cycles on the matrix pipe) needs its own 1 KiB weight fragment, i.e. one
// ds_read_b128 per 32 cycles: left to hipcc this came out as read -> s_waitcnt lgkmcnt(0..1) -> MFMA with one or two reads in
// flight, and the chains ran at LDS LATENCY, ~3 x the pipe's pace (deleting one of the three chains of the edge kernel took a
// third off the whole kernel: 0.92 -> 0.61 ms).  Here the fragments go through a register ring MGN_BF_RING deep, pinned by
// scheduling fences: MGN_BF_RING reads are in flight ahead of every MFMA.
#ifndef MGN_BF_RING
#define MGN_BF_RING 6
#endif
DEVINL void bfq_chunk(f32x16 (&acc)[4], const u32x4 (&in)[8], const bf16x8* w, int lane) {
    constexpr int D = MGN_BF_RING;
    const bf16x8* wl = w + lane;
    bf16x8 ring[D];
#pragma unroll
    for (int p = 0; p < D; ++p) ring[p] = wl[p * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const bf16x8 a = ring[i % D];
        if (i + D < 32) ring[i % D] = wl[(i + D) * 64];
        __builtin_amdgcn_sched_barrier(0);
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, in[i >> 2]), acc[i & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc = float(x) (no accumulate): the two bf16 of a dword are the high halves of two fp32 patterns
DEVINL void bfq_unpack(f32x16 (&acc)[4], const u32x4 (&x)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned w = x[s][d];
            const unsigned lo = w << 16, hi = w & 0xFFFF0000u;
            acc[s >> 1][8 * (s & 1) + 2 * d] = __builtin_bit_cast(float, lo);
            acc[s >> 1][8 * (s & 1) + 2 * d + 1] = __builtin_bit_cast(float, hi);
        }
}

// LayerNorm with packed fp32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma_f32 on register pairs): the bf16 kernels are bound by
// VALU issue, not by the MFMA pipe (where packing the fp32 kernel's LayerNorm cost time), so half the instructions pay.
DEVINL void layer_norm_frag_pk(f32x16 (&x)[4], const float* gamma, const float* beta, int h) {
    constexpr float invL = 1.0f / 128;
    f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            f32x2 v = {x[t][k], x[t][k + 1]};
            s2 += v;
        }
    float s = s2[0] + s2[1];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * invL;
    const f32x2 m2 = {mean, mean};
    f32x2 q2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            f32x2 v = {x[t][k], x[t][k + 1]};
            v -= m2;
            q2 = __builtin_elementwise_fma(v, v, q2);
            x[t][k] = v[0];
            x[t][k + 1] = v[1];
        }
    float q = q2[0] + q2[1];
    q += __shfl_xor(q, 32, 64);
    const float rstd = ln_rstd(q * invL, gamma, 128);
    const f32x2 r2 = {rstd, rstd};
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma) + h;
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta) + h;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 gv = g4[2 * (4 * t + g)];
            const f32x4 bv = b4[2 * (4 * t + g)];
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                f32x2 v = {x[t][4 * g + i], x[t][4 * g + i + 1]};
                const f32x2 gg = {gv[i], gv[i + 1]}, bb = {bv[i], bv[i + 1]};
                v = __builtin_elementwise_fma(v * r2, gg, bb);
                x[t][4 * g + i] = v[0];
                x[t][4 * g + i + 1] = v[1];
            }
        }
}

struct BfEdgeCtx {
    const bf16x8 *w1, *w2, *w3;
    const float* tb;
    BfSel sel;
    int lane0;
};
// one tile; xc: this tile's e rows (requested one tile ago), xn: the buffer the next tile's are requested into (at the TOP: a
// full tile of lead time for the HBM stream); the caller alternates the two buffers, so nothing is copied
DEVINL void bf_edge_tile(const BfEdgeArgs& a, const BfEdgeCtx& cx, int tile, int tile_next, int tile_next2, bool more, f32x16 (&acc)[4],
                         EdgeIdx& ix, EdgeIdx& ixn, u32x4 (&xc)[8], u32x4 (&xn)[8], int stamp_tile, int wave) {
    constexpr int L = 128;
    const int lane0 = cx.lane0;
    OPAQUE_LANE();
    STAMP(0);
    const bool valid = ix.r >= 0;
    const int r = valid ? ix.r : 0;
    u32x4 in[8], pn[8], qn[8];
#ifdef MGN_EXP_NOGATHER
    bfq_load<false>(pn, bfq_row_ptr(a.P, c, h), BF_STRIDE_ROW);
#else
    bfq_load<false>(pn, bfq_row_ptr(a.P, ixn.s, h), BF_STRIDE_ROW);
#endif
    bfq_load<true>(xn, bfq_tile_ptr(a.Elat, tile_next, lane), BF_STRIDE_TILE);
    const EdgeIdx ixnn = load_edge_idx_nb(a.snd, a.rcv, a.E, tile_next2, c);
    PHASE_FENCE();
    STAMP(1);
    bfq_chunk(acc, xc, cx.w1, lane);                             // layer 1 (edge part; P, Q, b1 are in acc)
    STAMP(2);
    bfq_pack<true>(in, acc);
    PHASE_FENCE();      // the bias table must not be read into 64 NEW registers while the old accumulator is still being packed
    tab_frag<4>(acc, cx.tb + T_B2 * L, h);
#ifndef MGN_EXP_NOL2
    bfq_chunk(acc, in, cx.w2, lane);                             // layer 2
#endif
    STAMP(3);
    bfq_pack<true>(in, acc);
    PHASE_FENCE();
    tab_frag<4>(acc, cx.tb + T_B3 * L, h);
    bfq_chunk(acc, in, cx.w3, lane);                             // layer 3
    STAMP(4);
    layer_norm_frag_pk(acc, cx.tb + T_GAMMA * L, cx.tb + T_BETA * L, h);   // acc = e' (fp32)
    // residual in fp32 (x + e', one rounding to bf16), stored non-temporally
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned w = xc[s][d];
            const bf16x2 v = __builtin_bit_cast(bf16x2, w);
            const int k = 8 * (s & 1) + 2 * d;
            in[s][d] = bf_pk2(__builtin_amdgcn_fdot2_f32_bf16(v, cx.sel.lo, acc[s >> 1][k], false),
                              __builtin_amdgcn_fdot2_f32_bf16(v, cx.sel.hi, acc[s >> 1][k + 1], false));
        }
    if (valid) bfq_store<true>(bfq_tile_ptr(a.Elat, tile, lane), BF_STRIDE_TILE, in);
    PHASE_FENCE();
    STAMP(5);
#ifdef MGN_EXP_NOGATHER
    bfq_load<false>(qn, bfq_row_ptr(a.Q, c, h), BF_STRIDE_ROW);
#else
    bfq_load<false>(qn, bfq_row_ptr(a.Q, ixn.r >= 0 ? ixn.r : 0, h), BF_STRIDE_ROW);           // next tile's Q rows
#endif
    // segmented sum of e' (fp32) over runs of equal receiver
    const int reff = valid ? r : (-4 - c);
    const int rprev = __shfl_up(reff, 1, 32);
    const int rnext = __shfl_down(reff, 1, 32);
    const bool head = (c == 0) || (reff != rprev);
    const unsigned hm = (unsigned)__ballot(head);
    const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
    const int st_in = max(start, c & 16);
    const bool c1 = (c - 1 >= st_in), c2 = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
    const bool cxr = (c >= 16) && (start <= 15);
#ifndef MGN_EXP_NOSCAN
    segmented_scan<4, true>(acc, c1, c2, c4, c8, cxr);
#endif
    STAMP(6);
    const bool tail = valid && ((c == 31) || (reff != rnext));
    const int r_first = __builtin_amdgcn_readfirstlane(reff);
    const bool sl = (start == 0) && (ix.r_before == r_first);
    const bool sr = (c == 31) && (ix.r_after == reff);
    const bool to_carry = sl || sr;
    u32x4* dst = to_carry ? bfq_row_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), h) : bfq_tile_ptr(a.AGG, r >> 5, 32 * h + (r & 31));
    bfq_pack<false>(in, acc);
    if (tail) bfq_store<false>(dst, to_carry ? BF_STRIDE_ROW : BF_STRIDE_TILE, in);
    if (!more) return;
    PHASE_FENCE();
    STAMP(7);
    bfq_unpack(acc, pn);                                         // next tile: acc = P[s] + Q[r]
    bfq_acc_add<false>(acc, qn, cx.sel);
    ix = ixn;
    ixn = ixnn;
}

// Pipelined bf16 edge kernel.  Per tile t (one wave, 32 edges; `acc` enters holding P[s] + Q[r] of this tile, `xc` its e tile):
//   top     request P[s(t+1)] (gathered by sender), the e tile of t+1 (streams from HBM) and the indices of tile t+2
//   middle  three MFMA chains, LayerNorm, residual store
//           request Q[r(t+1)] (receiver-sorted edges share it: mostly cache hits)
//   bottom  segmented scan, aggregate stores, then acc = float(P[s(t+1)]) + Q[r(t+1)] for the next tile
// so no request is waited for where it is issued.  Two waves per SIMD (256 registers): acc 64 + e tile x 2 (64) + in 32 + p 32
// + q 32.
__global__ __launch_bounds__(512, 2) void k_edge_bf16_pipe(const BfEdgeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
#pragma unroll
    for (int r = 0; r < 3; ++r) copy_to_lds16(wl + r * BF_CH, a.chunk[r], BF_CH, a.ntiles <= 16 * 1024);
    float* tb = smem + 3 * BF_CH / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    BfEdgeCtx cx;
    cx.w2 = reinterpret_cast<const bf16x8*>(wl);
    cx.w3 = reinterpret_cast<const bf16x8*>(wl + BF_CH);
    cx.w1 = reinterpret_cast<const bf16x8*>(wl + 2 * BF_CH);
    cx.tb = tb;
    const int lane0 = threadIdx.x & 63;
    cx.lane0 = lane0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TileWalk tw(a.ntiles, wave);
    if (tw.tile >= tw.end) return;
    cx.sel = bf_selectors();
    const int last = tw.tile + ((tw.end - 1 - tw.tile) / tw.stride) * tw.stride;   // this wave's last tile
    // (a request past the wave's last tile harmlessly repeats the last one: no divergent control flow around the loads)
    auto clampt = [&](int t) { return a.tile0 + (t <= last ? t : last); };
    // pipeline prologue: operands of the first tile, indices of the second
    EdgeIdx ix = load_edge_idx_nb(a.snd, a.rcv, a.E, clampt(tw.tile), lane0 & 31);
    EdgeIdx ixn = load_edge_idx_nb(a.snd, a.rcv, a.E, clampt(tw.tile + tw.stride), lane0 & 31);
    u32x4 xa[8], xb[8];
    f32x16 acc[4];
    {
        u32x4 p0[8], q0[8];
        bfq_load<true>(xa, bfq_tile_ptr(a.Elat, clampt(tw.tile), lane0), BF_STRIDE_TILE);
        bfq_load<false>(p0, bfq_row_ptr(a.P, ix.s, lane0 >> 5), BF_STRIDE_ROW);
        bfq_load<false>(q0, bfq_row_ptr(a.Q, ix.r >= 0 ? ix.r : 0, lane0 >> 5), BF_STRIDE_ROW);
        bfq_unpack(acc, p0);
        bfq_acc_add<false>(acc, q0, cx.sel);
    }
    int stamp_tile = 0;
    for (int t = tw.tile;; t += 2 * tw.stride, stamp_tile += 2) {
        const bool more1 = t + tw.stride <= last;
        bf_edge_tile(a, cx, a.tile0 + t, clampt(t + tw.stride), clampt(t + 2 * tw.stride), more1, acc, ix, ixn, xa, xb, stamp_tile, wave);
        if (!more1) break;
        const bool more2 = t + 2 * tw.stride <= last;
        bf_edge_tile(a, cx, a.tile0 + t + tw.stride, clampt(t + 2 * tw.stride), clampt(t + 3 * tw.stride), more2, acc, ix, ixn, xb, xa,
                     stamp_tile + 1, wave);
        if (!more2) break;
    }
}

// bf16 twin of load_aggregate; `y` is scratch (carry rows are summed in fp32 and rounded once)
DEVINL void bf_load_aggregate(bf16x8 (&in)[8], f32x16 (&y)[4], const int32_t* __restrict__ rowptr, const uint16_t* AGG, const uint16_t* CARRY,
                              int64_t zero_row, int tile, int nn, bool valid, int lane, int h) {
    const int a0 = valid ? rowptr[nn] : 0, a1 = valid ? rowptr[nn + 1] : 0;
    const int T1 = a0 >> 5, T2 = (a1 - 1) >> 5;
    const int extra = (a1 > a0 && T2 > T1) ? (T2 - T1) : 0;
    const bool from_agg = (a1 > a0) && !extra;
    const bf16x8* src0 = from_agg ? bf_tile_ptr(AGG, tile, lane) : bf_row_ptr(CARRY, extra ? (int64_t)(2 * T1 + 1) : zero_row, h);
    bf_load(in, src0, from_agg ? BF_STRIDE_TILE : BF_STRIDE_ROW);
    if (__any(extra > 0)) {
        zero_frag<4>(y);
        bf_unpack_add(y, in);
        {   // the second carry row of a run that straddles one tile boundary (the common case) without a loop: the other lanes add the
            // zero row (tile_common.hpp: LOAD_AGGREGATE); hub nodes continue in the loop
            bf16x8 cr[8];
            bf_load(cr, bf_row_ptr(CARRY, extra >= 1 ? (int64_t)2 * (T1 + 1) : zero_row, h), BF_STRIDE_ROW);
            bf_unpack_add(y, cr);
        }
        if (__any(extra >= 2))
            for (int q = 2; __any(q <= extra); ++q)
                if (q <= extra) {
                    bf16x8 cr[8];
                    bf_load(cr, bf_row_ptr(CARRY, (int64_t)2 * (T1 + q), h), BF_STRIDE_ROW);
                    bf_unpack_add(y, cr);
                }
        bf_pack(in, y);
    }
}

// node MLP: chunk[0]=W2 [1]=W3 [2]=W1v [3]=W1a, all resident (128 KiB)
__global__ __launch_bounds__(512, 2) void k_node_bf16(const BfNodeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
#pragma unroll
    for (int r = 0; r < 4; ++r) copy_to_lds16(wl + r * BF_CH, a.chunk[r], BF_CH, a.ntiles <= 16 * 1024);
    float* tb = smem + 4 * BF_CH / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const bf16x8* w2 = reinterpret_cast<const bf16x8*>(wl);
    const bf16x8* w3 = reinterpret_cast<const bf16x8*>(wl + BF_CH);
    const bf16x8* w1v = reinterpret_cast<const bf16x8*>(wl + 2 * BF_CH);
    const bf16x8* w1a = reinterpret_cast<const bf16x8*>(wl + 3 * BF_CH);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (TileWalk tw(a.ntiles, wave); tw.tile < tw.end; tw.tile += tw.stride) {
        OPAQUE_LANE();
        const int tile = tw.tile;
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        bf16x8 v[8], in[8];
        f32x16 acc[4], y[4];
        bf16x8* vtile = bf_tile_ptr(a.V, tile, lane);
        bf_load(v, vtile, BF_STRIDE_TILE);
        bf_load_aggregate(in, y, a.rowptr, a.AGG, a.CARRY, a.zero_row, tile, nn, valid, lane, h);
        tab_frag<4>(acc, tb + T_B1 * L, h);
        bf_chunk(acc, v, w1v, lane);                             // layer 1, node part
        bf_chunk(acc, in, w1a, lane);                            // layer 1, aggregate part
        if (a.AGG2) {                                            // second edge set's aggregate (its chunk streams from L2)
            bf_load_aggregate(in, y, a.rowptr2, a.AGG2, a.CARRY2, a.zero_row2, tile, nn, valid, lane, h);
            bf_chunk(acc, in, reinterpret_cast<const bf16x8*>(a.chunk[6]), lane);
        }
        relu_frag<4>(acc);
        bf_pack(in, acc);
        tab_frag<4>(y, tb + T_B2 * L, h);
        bf_chunk(y, in, w2, lane);
        relu_frag<4>(y);
        bf_pack(in, y);
        tab_frag<4>(acc, tb + T_B3 * L, h);
        bf_chunk(acc, in, w3, lane);
        layer_norm_frag<4>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);
        bf_unpack_add(acc, v);                                   // v <- v + v'  (fp32 add, one rounding)
        bf_pack(in, acc);
        if (valid) bf_store(vtile, BF_STRIDE_TILE, in);
    }
}

// ---- second-generation bf16 node kernels: the treatment of k_edge_bf16_pipe (requests ahead of use, weight-fragment ring,
// packed ReLU, ONE fp32 accumulator array).  V is read and written once per step -> non-temporal.
struct BfAggReq {       // where a node's aggregate comes from (bf_load_aggregate's address logic)
    const u32x4* src0;
    int stride0, T1, extra;
};
DEVINL BfAggReq bf_agg_req(int a0, int a1, bool valid, const uint16_t* AGG, const uint16_t* CARRY, int64_t zero_row, int tile, int lane, int h) {
    BfAggReq q;
    q.T1 = a0 >> 5;
    const int T2 = (a1 - 1) >> 5;
    q.extra = (valid && a1 > a0 && T2 > q.T1) ? (T2 - q.T1) : 0;
    const bool from_agg = valid && (a1 > a0) && !q.extra;
    q.src0 = from_agg ? bfq_tile_ptr(AGG, tile, lane) : bfq_row_ptr(CARRY, q.extra ? (int64_t)(2 * q.T1 + 1) : zero_row, h);
    q.stride0 = from_agg ? BF_STRIDE_TILE : BF_STRIDE_ROW;
    return q;
}
// in <- the node's aggregate as a packed row: g0 = the AGG slot or the first carry row, g1 = the second carry row of a run that
// straddles two edge tiles (one node in five on a triangle mesh; zero row otherwise), further carry rows (a receiver with more
// than 32 incoming edges) are fetched here.  Summed in fp32, rounded once.
DEVINL void bf_agg_sum(u32x4 (&in)[8], const u32x4 (&g0)[8], const u32x4 (&g1)[8], const BfAggReq& q, const uint16_t* CARRY, const BfSel& sel, int h) {
    (void)sel;
    if (!__any(q.extra > 0)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) in[s] = g0[s];
        return;
    }
    const bool more = __any(q.extra > 1);
    // piece by piece (eight values at a time): no 64-register fp32 copy of the row is ever live beside the accumulator
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float t[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned w0 = g0[s][d], w1 = g1[s][d];
            t[2 * d] = __builtin_bit_cast(float, w0 << 16) + __builtin_bit_cast(float, w1 << 16);
            t[2 * d + 1] = __builtin_bit_cast(float, w0 & 0xFFFF0000u) + __builtin_bit_cast(float, w1 & 0xFFFF0000u);
        }
        if (more)
            for (int k = 2; __any(k <= q.extra); ++k)
                if (k <= q.extra) {
                    const u32x4 cr = bfq_row_ptr(CARRY, (int64_t)2 * (q.T1 + k), h)[s * BF_STRIDE_ROW];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned w = cr[d];
                        t[2 * d] += __builtin_bit_cast(float, w << 16);
                        t[2 * d + 1] += __builtin_bit_cast(float, w & 0xFFFF0000u);
                    }
                }
#pragma unroll
        for (int d = 0; d < 4; ++d) in[s][d] = bf_pk2(t[2 * d], t[2 * d + 1]);
    }
}

// node MLP: chunk[0]=W2 [1]=W3 [2]=W1v [3]=W1a, all resident (128 KiB)
__global__ __launch_bounds__(512, 2) void k_node_bf16_pipe(const BfNodeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
#pragma unroll
    for (int r = 0; r < 4; ++r) copy_to_lds16(wl + r * BF_CH, a.chunk[r], BF_CH, a.ntiles <= 16 * 1024);
    float* tb = smem + 4 * BF_CH / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const bf16x8* w2 = reinterpret_cast<const bf16x8*>(wl);
    const bf16x8* w3 = reinterpret_cast<const bf16x8*>(wl + BF_CH);
    const bf16x8* w1v = reinterpret_cast<const bf16x8*>(wl + 2 * BF_CH);
    const bf16x8* w1a = reinterpret_cast<const bf16x8*>(wl + 3 * BF_CH);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TileWalk tw(a.ntiles, wave);
    if (tw.tile >= tw.end) return;
    const BfSel sel = bf_selectors();
    const int last = tw.tile + ((tw.end - 1 - tw.tile) / tw.stride) * tw.stride;
    auto clampt = [&](int t) { return t <= last ? t : last; };
    auto rp = [&](int t, int c, int& a0, int& a1) {          // CSR bounds of node (t, c); branch-free (clamped address)
        const int n = t * TILE + c;
        const int nn = n < a.n ? n : a.n - 1;
        a0 = a.rowptr[nn];
        a1 = a.rowptr[nn + 1];
    };
    u32x4 v[8];
    bfq_load<true>(v, bfq_tile_ptr(a.V, tw.tile, lane0), BF_STRIDE_TILE);
    int a0, a1;
    rp(tw.tile, lane0 & 31, a0, a1);
    for (int t = tw.tile;; t += tw.stride) {
        OPAQUE_LANE();
        const int n = t * TILE + c;
        const bool valid = n < a.n;
        // this tile's aggregate rows (the CSR bounds came a tile ahead), the next tile's CSR bounds
        const BfAggReq q = bf_agg_req(a0, a1, valid, a.AGG, a.CARRY, a.zero_row, t, lane, h);
        u32x4 g0[8], g1[8], in[8];
        bfq_load<false>(g0, q.src0, q.stride0);
        bfq_load<false>(g1, bfq_row_ptr(a.CARRY, q.extra ? (int64_t)2 * (q.T1 + 1) : a.zero_row, h), BF_STRIDE_ROW);
        int b0, b1;
        rp(clampt(t + tw.stride), c, b0, b1);
        PHASE_FENCE();
        f32x16 acc[4];
        tab_frag<4>(acc, tb + T_B1 * L, h);
        bfq_chunk(acc, v, w1v, lane);                            // layer 1, node part (the aggregate rows arrive meanwhile)
        bf_agg_sum(in, g0, g1, q, a.CARRY, sel, h);
        PHASE_FENCE();
        bfq_chunk(acc, in, w1a, lane);                           // layer 1, aggregate part
        if (a.AGG2) {                                            // second edge set's aggregate (its chunk streams from L2)
            int c0, c1;
            const int nn = valid ? n : 0;
            c0 = a.rowptr2[nn];
            c1 = a.rowptr2[nn + 1];
            const BfAggReq q2 = bf_agg_req(c0, c1, valid, a.AGG2, a.CARRY2, a.zero_row2, t, lane, h);
            bfq_load<false>(g0, q2.src0, q2.stride0);
            bfq_load<false>(g1, bfq_row_ptr(a.CARRY2, q2.extra ? (int64_t)2 * (q2.T1 + 1) : a.zero_row2, h), BF_STRIDE_ROW);
            bf_agg_sum(in, g0, g1, q2, a.CARRY2, sel, h);
            const bf16x8* w6 = reinterpret_cast<const bf16x8*>(a.chunk[6]);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w6[(s * 4 + tt) * 64 + lane], __builtin_bit_cast(bf16x8, in[s]), acc[tt], 0, 0, 0);
        }
        bfq_pack<true>(in, acc);
        PHASE_FENCE();
        tab_frag<4>(acc, tb + T_B2 * L, h);
        bfq_chunk(acc, in, w2, lane);                            // layer 2
        bfq_pack<true>(in, acc);
        PHASE_FENCE();
        tab_frag<4>(acc, tb + T_B3 * L, h);
        bfq_chunk(acc, in, w3, lane);                            // layer 3
        layer_norm_frag<4>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);
        bfq_acc_add<false>(acc, v, sel);                         // v <- v + v'  (fp32 add, one rounding)
        bfq_pack<false>(in, acc);
        if (valid) bfq_store<true>(bfq_tile_ptr(a.V, t, lane), BF_STRIDE_TILE, in);
        if (t + tw.stride > last) break;
        PHASE_FENCE();
        bfq_load<true>(v, bfq_tile_ptr(a.V, t + tw.stride, lane), BF_STRIDE_TILE);     // next tile's rows (v is dead)
        a0 = b0;
        a1 = b1;
    }
}

// P,Q projection, pipelined: chunk[4]=WP chunk[5]=WQ resident; the next tile's V rows are requested before this tile's chains
__global__ __launch_bounds__(512, 2) void k_project_bf16_pipe(const BfNodeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    copy_to_lds16(wl, a.chunk[4], BF_CH, a.ntiles <= 16 * 1024);
    copy_to_lds16(wl + BF_CH, a.chunk[5], BF_CH, a.ntiles <= 16 * 1024);
    float* tb = smem + 2 * BF_CH / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(wl);
    const bf16x8* wq = reinterpret_cast<const bf16x8*>(wl + BF_CH);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TileWalk tw(a.ntiles, wave);
    if (tw.tile >= tw.end) return;
    const int last = tw.tile + ((tw.end - 1 - tw.tile) / tw.stride) * tw.stride;
    u32x4 va[8], vb[8];
    bfq_load<false>(va, bfq_tile_ptr(a.V, a.tile0 + tw.tile, lane0), BF_STRIDE_TILE);
    auto body = [&](int t, u32x4 (&v)[8], u32x4 (&vn)[8]) {
        OPAQUE_LANE();
        const int tile = a.tile0 + t;
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        bfq_load<false>(vn, bfq_tile_ptr(a.V, a.tile0 + (t + tw.stride <= last ? t + tw.stride : last), lane), BF_STRIDE_TILE);
        PHASE_FENCE();
        u32x4 out[8];
        f32x16 acc[4];
        zero_frag<4>(acc);
        bfq_chunk(acc, v, wp, lane);
        bfq_pack<false>(out, acc);
        if (valid) bfq_store<false>(bfq_row_ptr(a.P, nn, h), BF_STRIDE_ROW, out);
        PHASE_FENCE();
        tab_frag<4>(acc, tb + T_BQ * L, h);
        bfq_chunk(acc, v, wq, lane);
        bfq_pack<false>(out, acc);
        if (valid) bfq_store<false>(bfq_row_ptr(a.Q, nn, h), BF_STRIDE_ROW, out);
    };
    for (int t = tw.tile;; t += 2 * tw.stride) {
        body(t, va, vb);
        if (t + tw.stride > last) break;
        body(t + tw.stride, vb, va);
        if (t + 2 * tw.stride > last) break;
    }
}

// P,Q projection: chunk[4]=WP chunk[5]=WQ resident
__global__ __launch_bounds__(512, 2) void k_project_bf16(const BfNodeArgs a) {
    constexpr int L = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    copy_to_lds16(wl, a.chunk[4], BF_CH, a.ntiles <= 16 * 1024);
    copy_to_lds16(wl + BF_CH, a.chunk[5], BF_CH, a.ntiles <= 16 * 1024);
    float* tb = smem + 2 * BF_CH / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(wl);
    const bf16x8* wq = reinterpret_cast<const bf16x8*>(wl + BF_CH);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (TileWalk tw(a.ntiles, wave); tw.tile < tw.end; tw.tile += tw.stride) {
        OPAQUE_LANE();
        const int tile = a.tile0 + tw.tile;
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        bf16x8 v[8], out[8];
        f32x16 acc[4];
        bf_load(v, bf_tile_ptr(a.V, tile, lane), BF_STRIDE_TILE);
        zero_frag<4>(acc);
        bf_chunk(acc, v, wp, lane);
        bf_pack(out, acc);
        if (valid) bf_store(bf_row_ptr(a.P, nn, h), BF_STRIDE_ROW, out);
        tab_frag<4>(acc, tb + T_BQ * L, h);
        bf_chunk(acc, v, wq, lane);
        bf_pack(out, acc);
        if (valid) bf_store(bf_row_ptr(a.Q, nn, h), BF_STRIDE_ROW, out);
    }
}

// fp32 tile-major piece m = 4t+g (features 32t+8g+4h+i)  <->  bf16 piece s (features f(s,h,j)): bf16 piece s of a
// lane is the concatenation of its fp32 pieces 2s and 2s+1.
__global__ void k_tile_f32_to_bf16(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n_pieces) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one bf16 piece (tile, s, lane)
    if (i >= n_pieces) return;
    const int64_t tile = i / 512;
    const int rem = (int)(i - tile * 512), sidx = rem >> 6, lane = rem & 63;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src) + tile * 1024 + lane;
    const f32x4 lo = s4[(2 * sidx) * 64], hi = s4[(2 * sidx + 1) * 64];
    bf16x8 o;
    o[0] = (__bf16)lo[0]; o[1] = (__bf16)lo[1]; o[2] = (__bf16)lo[2]; o[3] = (__bf16)lo[3];
    o[4] = (__bf16)hi[0]; o[5] = (__bf16)hi[1]; o[6] = (__bf16)hi[2]; o[7] = (__bf16)hi[3];
    reinterpret_cast<bf16x8*>(dst)[i] = o;
}
__global__ void k_tile_bf16_to_f32(const uint16_t* __restrict__ src, float* __restrict__ dst, int64_t n_pieces) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pieces) return;
    const int64_t tile = i / 512;
    const int rem = (int)(i - tile * 512), sidx = rem >> 6, lane = rem & 63;
    const bf16x8 v = reinterpret_cast<const bf16x8*>(src)[i];
    f32x4* d4 = reinterpret_cast<f32x4*>(dst) + tile * 1024 + lane;
    f32x4 lo, hi;
    lo[0] = (float)v[0]; lo[1] = (float)v[1]; lo[2] = (float)v[2]; lo[3] = (float)v[3];
    hi[0] = (float)v[4]; hi[1] = (float)v[5]; hi[2] = (float)v[6]; hi[3] = (float)v[7];
    d4[(2 * sidx) * 64] = lo;
    d4[(2 * sidx + 1) * 64] = hi;
}
// dst row stride `dst4` in 16-byte pieces (> row width when the rows of several edge sets interleave in a halo message)
__global__ void k_gather_rows16(const uint16_t* __restrict__ src, const int32_t* __restrict__ idx, uint16_t* __restrict__ dst, int64_t rows,
                                int dst4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte piece; 16 per row
    if (i >= rows * 16) return;
    const int64_t r = i >> 4;
    reinterpret_cast<f32x4*>(dst)[r * dst4 + (i & 15)] = reinterpret_cast<const f32x4*>(src)[bf_prow_piece(idx[r], (int)(i & 15))];
}
// halo unpack, bf16: plain rows of the receive buffer -> rows row0 .. of bP
__global__ void k_scatter_prows16(const uint16_t* __restrict__ src, int src4, uint16_t* __restrict__ dst, int64_t row0, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 16) return;
    const int64_t r = i >> 4;
    reinterpret_cast<f32x4*>(dst)[bf_prow_piece(row0 + r, (int)(i & 15))] = reinterpret_cast<const f32x4*>(src)[r * src4 + (i & 15)];
}

// ================================================================================================
// small utility kernels
// ================================================================================================
__global__ void k_gather_rows(const float* __restrict__ src, const int32_t* __restrict__ idx, float* __restrict__ dst,
                              int64_t rows, int L4, int dst4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * L4) return;
    const int64_t r = i / L4;
    const int q = (int)(i - r * L4);
    reinterpret_cast<f32x4*>(dst)[r * dst4 + q] = reinterpret_cast<const f32x4*>(src)[prow_f4(idx[r], q, 4 * L4)];      // src: P rows (frag.hpp: prow_ptr)
}
// halo unpack: plain rows of the receive buffer -> rows row0 .. of P
__global__ void k_scatter_prows(const float* __restrict__ src, int src4, float* __restrict__ dst, int64_t row0, int64_t rows, int L4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * L4) return;
    const int64_t r = i / L4;
    const int q = (int)(i - r * L4);
    reinterpret_cast<f32x4*>(dst)[prow_f4(row0 + r, q, 4 * L4)] = reinterpret_cast<const f32x4*>(src)[r * src4 + q];
}

// caller-order row-major rows  <->  engine order, tile-major storage (mgn_latents_import / export on the device)
// one thread per 16-byte piece of a local row: tile-major piece m of lane (c,h) <- row-major float4 index 2m+h
__global__ void k_rows_to_tiles(const float* __restrict__ src, const int64_t* __restrict__ gid64, const int32_t* __restrict__ gid32,
                                float* __restrict__ dst, int64_t rows, int L) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pieces = L / 4;
    if (i >= rows * pieces) return;
    const int64_t r = i / pieces;
    const int q = (int)(i - r * pieces), m = q >> 1, hh = q & 1;
    const int64_t g = gid64 ? gid64[r] : (int64_t)gid32[r];
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[g * pieces + q];
    reinterpret_cast<f32x4*>(dst)[(r / TILE) * (TILE * pieces) + (int64_t)m * 64 + 32 * hh + (r % TILE)] = v;
}
__global__ void k_tiles_to_rows(const float* __restrict__ src, const int64_t* __restrict__ gid64, const int32_t* __restrict__ gid32,
                                float* __restrict__ dst, int64_t rows, int L) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pieces = L / 4;
    if (i >= rows * pieces) return;
    const int64_t r = i / pieces;
    const int q = (int)(i - r * pieces), m = q >> 1, hh = q & 1;
    const int64_t g = gid64 ? gid64[r] : (int64_t)gid32[r];
    reinterpret_cast<f32x4*>(dst)[g * pieces + q] =
        reinterpret_cast<const f32x4*>(src)[(r / TILE) * (TILE * pieces) + (int64_t)m * 64 + 32 * hh + (r % TILE)];
}

DEVINL uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// N(0,1) keyed by (seed, global row id, feature): identical whatever the partition (KAT-7 at scale).
// dst is TILE-MAJOR storage of `rows` rows padded to whole 32-row tiles; padding rows are zeroed.
__global__ void k_randn_rows(float* __restrict__ dst, const int64_t* __restrict__ gid64, const int32_t* __restrict__ gid32,
                             int64_t rows, int L, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ntl = (rows + TILE - 1) / TILE;
    if (i >= ntl * TILE * L) return;
    const int64_t tile = i / (TILE * L);
    const int rem = (int)(i - tile * (TILE * L));
    const int m = rem >> 8, lane = (rem & 255) >> 2, q = rem & 3;
    const int64_t r = tile * TILE + (lane & 31);
    const int f = 32 * (m >> 2) + 8 * (m & 3) + 4 * (lane >> 5) + q;
    if (r >= rows) { dst[i] = 0.f; return; }
    const uint64_t g = gid64 ? (uint64_t)gid64[r] : (gid32 ? (uint64_t)gid32[r] : (uint64_t)r);
    const uint64_t bits = splitmix64(seed ^ splitmix64(g * (uint64_t)L + (uint64_t)f));
    const float u1 = ((float)(uint32_t)(bits >> 40) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
    const float u2 = (float)(uint32_t)((bits >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
    dst[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// deterministic: every block writes its own partial (fixed grid, fixed per-thread stride), the host adds
// the partials in block order.  out[2*block] = sum, out[2*block+1] = sum of squares.
constexpr int CHECKSUM_BLOCKS = 1024;
__global__ void k_checksum(const float* __restrict__ src, int64_t n, double* __restrict__ out) {
    __shared__ double sh[2][4];
    double s = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)src[i];
        s += v;
        q += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        q += __shfl_xor(q, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        out[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

// ---- ODE-driver helpers (native rollout) ------------------------------------------------
// out = u + dt * sum_j c[j] * k[j]   (n elements; up to 7 stages)
__global__ void k_lincomb(float* __restrict__ out, const float* __restrict__ u, LinComb lc, float dt, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j)
        if (j < lc.n) s = fmaf(lc.c[j], lc.k[j][i], s);
    out[i] = fmaf(dt, s, u[i]);
}

// x[n][:] = frame[n][:] where mask[n]   (ode_func_eval's in-place inflow overwrite, reference src/solve.jl:151-152)
__global__ void k_overwrite(float* __restrict__ x, const float* __restrict__ frame, const uint8_t* __restrict__ mask, int64_t N, int O) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * O) return;
    if (mask[i / O]) x[i] = frame[i];
}

// partial sums of ((dt * sum_j c[j] k[j]) / (atol + rtol * max(|u|, |unew|)))^2 : deterministic per-block partials
__global__ void k_errnorm(const float* __restrict__ u, const float* __restrict__ unew, LinComb lc, float dt, float atol,
                          float rtol, int64_t n, double* __restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (j < lc.n) s = fmaf(lc.c[j], lc.k[j][i], s);
        const float sc = atol + rtol * fmaxf(fabsf(u[i]), fabsf(unew[i]));
        const double r = (double)(dt * s) / (double)sc;
        acc += r * r;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

hipError_t launch_lincomb(float* out, const float* u, const LinComb& lc, float dt, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_lincomb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, u, lc, dt, n);
    return hipGetLastError();
}
hipError_t launch_overwrite(float* x, const float* frame, const uint8_t* mask, int64_t N, int O, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_overwrite, dim3((unsigned)((N * O + 255) / 256)), dim3(256), 0, s, x, frame, mask, N, O);
    return hipGetLastError();
}
int errnorm_partials() { return 64; }
hipError_t launch_errnorm(const float* u, const float* unew, const LinComb& lc, float dt, float atol, float rtol, int64_t n,
                          double* partial, hipStream_t s) {
    hipLaunchKernelGGL(k_errnorm, dim3(64), dim3(256), 0, s, u, unew, lc, dt, atol, rtol, n, partial);
    return hipGetLastError();
}

// ================================================================================================
// launch wrappers
// ================================================================================================
static int g_num_cu = 0;
static int num_cus() {
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            g_num_cu = p.multiProcessorCount;
        else
            g_num_cu = 256;
    }
    return g_num_cu;
}

constexpr size_t LDS_BYTES = 160 * 1024;

// resident chunks for L: chunk bytes = L*L*4; tables T_COUNT*L*4
static int resident_chunks(int L, int want) {
    const size_t ch = (size_t)L * L * 4, tabs = (size_t)T_COUNT * L * 4 + 64;
    int r = (int)((LDS_BYTES - tabs) / ch);
    return r < want ? r : want;
}

// Launches with at most one tile per SIMD cannot amortise a per-block LDS weight preload (up to 156 KB copied by
// 64-128 threads = 10-17 us): such launches use the all-streaming instantiations (weights stay in L2).
// g_path: 0 auto, 1 force the LDS-resident persistent kernels, 2 force all-streaming, 3 force cooperative
// (tests exercise every path on small graphs through mgn_debug_kernel_path)
static int g_path = [] { const char* e = getenv("MGN_KERNEL_PATH"); return e ? atoi(e) : 0; }();   // experiments
int set_kernel_path(int p) { const int old = g_path; g_path = p; return old; }
static int g_c16_rt = [] { const char* e = getenv("MGN_C16_RT"); return e ? atoi(e) : 0; }();   // 0: by size; 1..3: 16-edge tiles per block
// Large fp32 launches at L = 128, hidden_layers = 2 run on the bf16 matrix cores with every operand split exactly into three bf16
// pieces (split.hip / split_ws.hip: fp32 storage, fp32 accumulation, error against float64 no worse than the fp32-MFMA kernels').
// MGN_FP32_SPLIT: 0 = the fp32-MFMA kernels (v_mfma_f32_32x32x2_f32: the fp32 reference path), 1 = default (edge step: the
// lock-step kernel with the shared LDS weight ring), 2 = the edge step with per-wave register rings (A/B).
static int g_fp32_split = [] { const char* e = getenv("MGN_FP32_SPLIT"); return e ? atoi(e) : 1; }();
int set_fp32_split(int on) { const int old = g_fp32_split; g_fp32_split = on; return old; }
int fp32_split_enabled() { return g_fp32_split; }
// 16-row cooperative kernels on the split path: bit 0 the edge kernel (with two or three row tiles per block its chains are
// matrix-bound on the fp32 pipe: 17.2 -> 14.1 us on the cylinder mesh), bit 1 the node kernel (one row tile per block streams 96 KiB of
// weight pieces per chunk where the fp32 fragments are 64 and is bound by that stream either way: 11.6 -> 11.3 us), bit 2 the edge
// kernel at one row tile per block as well.  Cylinder mesh, per processor step: 29.3 us (0), 26.1 (1), 25.8 (3, the default).
static int g_c16_split = [] { const char* e = getenv("MGN_C16_SPLIT"); return e ? atoi(e) : 3; }();
int set_c16_split(int on) { const int old = g_c16_split; g_c16_split = on; return old; }
int c16_split_enabled() { return g_c16_split; }
// two fp16 pieces and three products instead of three bf16 pieces and six (k_edge_ring_h, split.hip); MGN_SPLIT_F16=0: the bf16 pieces
static int g_split_f16 = [] { const char* e = getenv("MGN_SPLIT_F16"); return e ? atoi(e) : 1; }();
int set_split_f16(int on) { const int old = g_split_f16; g_split_f16 = on; return old; }
int split_f16_enabled() { return g_split_f16; }
int set_c16_row_tiles(int rt) { const int old = g_c16_rt; g_c16_rt = rt; return old; }
int get_kernel_path() { return g_path; }
static bool small_launch(int ntiles) { return g_path == 0 ? ntiles <= 4 * num_cus() : g_path >= 2; }
// cooperative (4 waves per tile) kernels: up to this many tiles per CU for the edge / node kernels (size sweep)
static int g_tail_coop = [] { const char* e = getenv("MGN_TAIL_COOP"); return e ? atoi(e) : 1; }();   // 0: whole launch persistent
static int g_coop_edge = [] { const char* e = getenv("MGN_COOP_EDGE_TILES_PER_CU"); return e ? atoi(e) : 16; }();
static int g_coop_node = [] { const char* e = getenv("MGN_COOP_NODE_TILES_PER_CU"); return e ? atoi(e) : 8; }();
static int g_coop16 = [] { const char* e = getenv("MGN_COOP16"); return e ? atoi(e) : 1; }();   // 16-row cooperative tiles on small graphs
int coop16_enabled() { return (g_coop16 && (g_path == 0 || g_path == 5)) ? 1 : 0; }
// the 16-row tiles pay while the launches are latency-bound: up to this many 32-row tiles per CU (size sweep, docs/experiments.md)
// (`ring_hs`: the handle's large-mesh edge kernel would be k_edge_ring_hs -- fp32, one edge set, two fp16 pieces -- whose 28 KiB LDS prologue
// lets it take over a tile per CU earlier: 3 025 nodes 38 -> 35 us per step, 4 096 nodes 42 -> 37)
static int g_c16_edge = [] { const char* e = getenv("MGN_C16_EDGE_TILES_PER_CU"); return e ? atoi(e) : 0; }();   // 0: 3, or 2 with ring_hs
static int g_c16_node = [] { const char* e = getenv("MGN_C16_NODE_TILES_PER_CU"); return e ? atoi(e) : 1; }();
bool coop16_size(int ntiles_e, int ntiles_n, bool ring_hs) {
    if (g_path == 5) return true;
    const int lim = g_c16_edge > 0 ? g_c16_edge : (ring_hs ? 2 : 3);
    return ntiles_e <= lim * num_cus() && ntiles_n <= g_c16_node * num_cus();
}
int set_c16_edge_tiles(int t) { const int old = g_c16_edge; g_c16_edge = t; return old; }   // (tests: 0 = by the handle, n = n tiles per CU)
bool ring_hs_default() { return g_fp32_split == 1 && g_split_f16 != 0 && g_path == 0 && edge_ring_h_streamed() != 0; }
static bool coop_size(int ntiles, bool edge) { return g_path == 0 ? ntiles <= (edge ? g_coop_edge : g_coop_node) * num_cus() : (g_path == 3 || g_path == 5); }
bool launch_is_small(int ntiles) { return coop_size(ntiles, false); }
// where the split-path node kernels (k_node_split + k_project_split) take the node side from the cooperative tiles: above two tiles per CU
// (19.6 k nodes: 50 -> 43 us, 32 k: 66 -> 50, 62 k: 131 -> 98; at 16 k, two tiles per CU, the cooperative kernels are faster: 37 vs 48)
static int g_node16_mid = [] { const char* e = getenv("MGN_NODE16_MID"); return e ? atoi(e) : 1; }();   // 0: cooperative 32-row node kernel (fp32 pipe) there
static const int g_node_split_min = [] { const char* e = getenv("MGN_NODE_SPLIT_MIN_TILES_PER_CU"); return e ? atoi(e) : 2; }();
bool node_split_size(int ntiles) { return g_fp32_split != 0 && g_path == 0 && ntiles > g_node_split_min * num_cus(); }
bool launch_is_small_edge(int ntiles_e) { return coop_size(ntiles_e, true); }


static LaunchCfg tile_launch(int L, int ntiles, int nres) {
    LaunchCfg lc;
    const int cus = num_cus();
    int wpb = 8;
    if (ntiles <= cus) wpb = 1;
    else if (ntiles <= 2 * cus) wpb = 2;
    else if (ntiles <= 4 * cus) wpb = 4;
    int blocks = (ntiles + wpb - 1) / wpb;
    if (blocks > cus) blocks = cus;
    blocks = ((blocks + NUM_XCD - 1) / NUM_XCD) * NUM_XCD;
    lc.blocks = blocks;
    lc.threads = wpb * 64;
    lc.lds = (size_t)nres * L * L * 4 + (size_t)T_COUNT * L * 4 + 64;  // + spare
    return lc;
}

template <typename K, typename A>
static hipError_t launch_k(K kern, const A& a, const LaunchCfg& lc, hipStream_t s) {
    // opt in to > 64 KiB dynamic LDS once per kernel and size (small meshes are launch-bound: keep this off the
    // per-launch path)
    {
        static std::mutex mu;
        static std::unordered_map<const void*, size_t> granted;   // keyed by kernel: K is only the signature type
        std::lock_guard<std::mutex> lock(mu);
        size_t& g = granted[reinterpret_cast<const void*>(kern)];
        if (lc.lds > g) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lc.lds);
            if (e != hipSuccess) return e;
            g = lc.lds;
        }
    }
    hipLaunchKernelGGL(kern, dim3(lc.blocks), dim3(lc.threads), lc.lds, s, a);
    return hipGetLastError();
}

#define DISPATCH_L(KERN, WANT, ARGS, NTILES)                                                   \
    do {                                                                                       \
        if ((NTILES) <= 0) return hipSuccess;                                                  \
        const int nres = resident_chunks(L, WANT);                                             \
        const LaunchCfg lc = tile_launch(L, NTILES, nres);                                     \
        const bool h2_ = L == 128 && g_fp32_split && g_split_f16 && (ARGS).splith[0];          \
        if (L == 128 && small_launch(NTILES)) {                                                \
            LaunchCfg l0 = lc;                                                                 \
            l0.lds = (size_t)T_COUNT * L * 4 + 64;                                             \
            return h2_ ? launch_k(KERN<4, 0, false, true>, ARGS, l0, s) : launch_k(KERN<4, 0>, ARGS, l0, s); \
        }                                                                                      \
        if (L == 128 && h2_) return launch_k(KERN<4, (WANT < 2 ? WANT : 2), false, true>, ARGS, lc, s); \
        if (L == 128) return launch_k(KERN<4, (WANT < 2 ? WANT : 2)>, ARGS, lc, s);            \
        if (L == 64) return launch_k(KERN<2, WANT>, ARGS, lc, s);                              \
        if (L == 32) return launch_k(KERN<1, WANT>, ARGS, lc, s);                              \
        return hipErrorInvalidValue;                                                           \
    } while (0)

hipError_t launch_project(int L, const NodeArgs& a, hipStream_t s);

// hidden_layers != 2 (GenMlp): the GEN instantiations, every weight chunk streamed from L2, tables only in LDS
static LaunchCfg gen_launch(int L, int ntiles) {
    LaunchCfg lc = tile_launch(L, ntiles, 0);
    lc.lds = (size_t)T_COUNT * L * 4 + 64;
    return lc;
}
#define DISPATCH_GEN(L_, KERN4, KERN2, KERN1, ARGS, NTILES)                          \
    do {                                                                             \
        if ((NTILES) <= 0) return hipSuccess;                                        \
        const LaunchCfg lg = gen_launch(L_, NTILES);                                 \
        if ((L_) == 128) return launch_k(KERN4, ARGS, lg, s);                        \
        if ((L_) == 64) return launch_k(KERN2, ARGS, lg, s);                         \
        if ((L_) == 32) return launch_k(KERN1, ARGS, lg, s);                         \
        return hipErrorInvalidValue;                                                 \
    } while (0)

// pinned weight rings (coop_chain_primed<true>) while a launch has at most MGN_COOP_FENCE_TILES_PER_CU (default 4) tiles per CU
static bool coop_fence(int ntiles) {
    static const int per_cu = [] { const char* e = getenv("MGN_COOP_FENCE_TILES_PER_CU"); return e ? atoi(e) : 4; }();
    return ntiles <= per_cu * 256;
}
static size_t coop_lds() { return (size_t)2 * 16 * 64 * 16 + (size_t)T_COUNT * 128 * 4; }
static bool coop_ok(int L, int ntiles, const float* const* chunk_t, bool edge = false) {
    return L == 128 && chunk_t[0] != nullptr && coop_size(ntiles, edge);
}

// which kernel family the last fp32 edge launch went to (bench.py labels its roofline with what RAN, not with the global switches):
// 1 generic hidden_layers, 2 16-row small-graph, 3 cooperative 4-wave tiles, 4 all-streaming, (5, 6: kernels retired in round 5,)
// 7 k_edge_ring<8>, 8 k_edge_ring<4>, 9 k_edge_step<4,2> (fp32-MFMA persistent), (10, 11: retired,) 12 k_edge_coop16m on the
// split path, 13 / 14 k_edge_ring_h<8 / 4> (two fp16 pieces: the default of large fp32 launches)
static int g_last_edge_kernel = 0;
int last_edge_kernel() { return g_last_edge_kernel; }
// the family of the launch that carries a set's edges: a launch over less than half of the set's tiles (the boundary tiles of a partitioned pass,
// launched AFTER the interior ones) does not speak for the step
#define SET_LAST_EDGE(A, CODE)                                                            \
    do {                                                                                  \
        if (2 * (int64_t)(A).ntiles * TILE >= (A).E) g_last_edge_kernel = (CODE);         \
    } while (0)

hipError_t launch_edge_step(int L, const EdgeArgs& a, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    if (a.gen.use && a.ElatSrc) return hipErrorInvalidValue;
    if (a.gen.use) SET_LAST_EDGE(a, 1);
    if (a.gen.use) DISPATCH_GEN(L, (k_edge_step<4, 0, true>), (k_edge_step<2, 0, true>), (k_edge_step<1, 0, true>), a, a.ntiles);
    const int nres = resident_chunks(L, 3);
    LaunchCfg lc = tile_launch(L, a.ntiles, nres);
    if (a.c16 && L == 128 && a.chunk_t[0]) {        // small graph: 16-row tiles, 4 waves each (the handle decided for both kernels)
        // RT 16-edge tiles per block: about one block per CU (MGN_C16_RT = 1..3 pins it)
        SET_LAST_EDGE(a, 2);
        const int nht = 2 * a.ntiles;
        int rt = g_c16_rt ? g_c16_rt : (nht + num_cus() - 1) / num_cus();
        const bool sp16 = g_fp32_split && (g_c16_split & 1) && a.split16[0];
        // (four to six row tiles per block exist on the split path only: one round of blocks up to 24.5 k edges instead of a second,
        // partly filled one behind three tiles per block -- 12.5 k edges: 35.4 -> 2x us per step)
        const int rt_max = (sp16 && !a.bf && !g_c16_rt) ? 6 : 3;
        rt = rt < 1 ? 1 : (rt > rt_max ? rt_max : rt);
        LaunchCfg c16{(nht + rt - 1) / rt, 256, (size_t)rt * 2 * 8 * 64 * 16 + (size_t)rt * 2 * 64 * 4};
        if (g_fp32_split && (g_c16_split & 1) && a.split16[0] && (rt >= 2 || (g_c16_split & 4))) {   // split path: bf16 matrix cores at fp32 accuracy (pieces exchanged: 12 KiB per tile)
            c16.lds = (size_t)rt * 2 * 12 * 64 * 16 + (size_t)rt * 2 * 64 * 4;
            const bool h2 = g_split_f16 && a.split16h[0] && rt <= 4;   // two fp16 pieces, three products (else three bf16 pieces, six; five and six row tiles spill with two)
            SET_LAST_EDGE(a, h2 ? 15 : 12);
#define C16E(RT_, BF_) (h2 ? launch_k(k_edge_coop16m<RT_, BF_, 2>, a, c16, s) : launch_k(k_edge_coop16m<RT_, BF_, 1>, a, c16, s))
            if (a.bf) {
                if (rt == 3) return C16E(3, true);
                if (rt == 2) return C16E(2, true);
                return C16E(1, true);
            }
            if (rt == 6) return C16E(6, false);
            if (rt == 5) return C16E(5, false);
            if (rt == 4) return C16E(4, false);
            if (rt == 3) return C16E(3, false);
            if (rt == 2) return C16E(2, false);
            return C16E(1, false);
#undef C16E
        }
        if (a.bf) {
            if (rt == 3) return launch_k(k_edge_coop16m<3, true>, a, c16, s);
            if (rt == 2) return launch_k(k_edge_coop16m<2, true>, a, c16, s);
            return launch_k(k_edge_coop16m<1, true>, a, c16, s);
        }
        if (rt == 3) return launch_k(k_edge_coop16m<3, false>, a, c16, s);
        if (rt == 2) return launch_k(k_edge_coop16m<2, false>, a, c16, s);
        return launch_k(k_edge_coop16m<1, false>, a, c16, s);
    }
    if (a.ElatSrc) return hipErrorInvalidValue;      // only the 16-row kernel reads its e rows from a second array
    // where the ring kernel of the split path is available it takes over from the cooperative tiles at 3 tiles per CU already
    // (four-wave blocks; 5 k nodes: 50 vs 68 us per step, 10 k: 85 vs 110, 16 k: 110 vs 146), the fp32-MFMA persistent kernels only at 16
    static const int coop_edge_ring_env = [] { const char* e = getenv("MGN_COOP_EDGE_TILES_PER_CU_RING"); return e ? atoi(e) : 0; }();
    const bool ring_ok = L == 128 && g_fp32_split == 1 && a.split[0] && g_path == 0;
    const int coop_edge_ring = coop_edge_ring_env > 0 ? coop_edge_ring_env : ((ring_ok && g_split_f16 && a.splith[0] && edge_ring_h_streamed()) ? 2 : 3);
    if (coop_ok(L, a.ntiles, a.chunk_t, true) && !(ring_ok && a.ntiles > coop_edge_ring * num_cus())) {   // small graph: 4 waves per tile
        SET_LAST_EDGE(a, 3);
        LaunchCfg c4{a.ntiles, 256, coop_lds()};
        return coop_fence(a.ntiles) ? launch_k(k_edge_coop<true>, a, c4, s) : launch_k(k_edge_coop<false>, a, c4, s);
    }
    if (L == 128) {
        if (small_launch(a.ntiles) && !(ring_ok && a.ntiles > coop_edge_ring * num_cus())) {   // few tiles: the per-block LDS preload would dominate -> stream everything from L2
            SET_LAST_EDGE(a, 4);
            lc.lds = (size_t)T_COUNT * L * 4 + 64;
            return launch_k(k_edge_step<4, 0>, a, lc, s);
        }
        if (g_fp32_split && a.split[0] && g_path == 0) {   // split path (split.hip): fp32 accuracy on the bf16 matrix cores
            LaunchCfg ls = lc;
            ls.lds = (size_t)3 * 32768 + (size_t)3 * 16384 + (size_t)T_COUNT * L * 4 + 64;
            static const int ring_waves = [] { const char* e = getenv("MGN_RING_WAVES"); return e ? atoi(e) : 0; }();   // 0: by size
            // four-wave blocks (one wave per SIMD) up to 2.5 rounds of eight-wave blocks: 16 k nodes 75 vs 83 us, 25.6 k 115 vs 120,
            // 40 k 175 vs 160 (docs/experiments.md)
            // k_edge_ring_hs (28 KiB of LDS prologue per block instead of 150): a round of four-wave blocks takes ~0.62 of a round of eight-wave
            // blocks for half the tiles, so the shape with the shorter sum of rounds runs -- four waves up to 4 and from 8 to 12 tiles per CU
            // (M-1M slices of 100 / 128 / 145 / 160 nodes a side: 34 vs 39, 57 vs 55, 59 vs 70, 79 vs 83 us for eight vs four waves)
            bool four = a.ntiles <= 20 * num_cus();
            if (g_split_f16 && a.splith[0] && edge_ring_h_streamed()) {
                const int r4 = (a.ntiles + 4 * num_cus() - 1) / (4 * num_cus()), r8 = (a.ntiles + 8 * num_cus() - 1) / (8 * num_cus());
                four = 0.62 * r4 < (double)r8;
            }
            if (ring_waves == 4 || (ring_waves == 0 && four)) {
                ls.threads = 256;
                // as few blocks as the number of rounds allows (every block pays the 150 KiB LDS prologue)
                const int rounds = (a.ntiles + 4 * num_cus() - 1) / (4 * num_cus());
                int blocks = (a.ntiles + 4 * rounds - 1) / (4 * rounds);
                if (blocks > num_cus()) blocks = num_cus();
                ls.blocks = ((blocks + NUM_XCD - 1) / NUM_XCD) * NUM_XCD;
            }
            if (g_split_f16 && a.splith[0]) {
                ls.lds = edge_ring_h_lds();
                SET_LAST_EDGE(a, edge_ring_h_streamed() ? (ls.threads == 256 ? 17 : 16) : (ls.threads == 256 ? 14 : 13));
                return launch_edge_ring_h(a, ls, s);
            }
            SET_LAST_EDGE(a, ls.threads == 256 ? 8 : 7);
            return launch_edge_ring(a, ls, s);
        }
        SET_LAST_EDGE(a, 9);
        lc.lds += (size_t)MGN_EDGE_JR * 64 * 4 * 4;   // partially resident third chunk
        // Tail of the persistent walk: with r = ntiles / (8 waves x 256 blocks) rounds, a last round that is less than
        // ~60 % full costs a whole tile-time (11.4 tiles per wave on an 8-GPU partition of M-1M: 5 %).  Those tiles go to
        // the cooperative kernel instead (4 waves per tile: a third of the latency), launched behind the persistent one.
        const int nw = lc.blocks * (lc.threads / 64);
        const int rem = a.ntiles % nw, rounds = a.ntiles / nw;
        // (Launches with fewer than MGN_SPREAD_ROUNDS rounds spread their last round over all CUs instead -- TileWalk -- and
        // are faster without the split: -2.7 % at 125 k nodes, the per-GPU share of M-1M on 8 GPUs.)
        if (g_tail_coop && g_path == 0 && a.chunk_t[0] && rounds >= MGN_SPREAD_ROUNDS && rem > 0 && rem * 10 < nw * 6) {
            EdgeArgs body = a, tail = a;
            body.ntiles = a.ntiles - rem;
            tail.tile0 = a.tile0 + body.ntiles;
            tail.ntiles = rem;
            const hipError_t e = launch_k(k_edge_step<4, 2>, body, lc, s);
            if (e != hipSuccess) return e;
            LaunchCfg c4{rem, 256, coop_lds()};
            return launch_k(k_edge_coop<false>, tail, c4, s);
        }
        return launch_k(k_edge_step<4, 2>, a, lc, s);
    }
    if (L == 64) return launch_k(k_edge_step<2, 3>, a, lc, s);
    if (L == 32) return launch_k(k_edge_step<1, 3>, a, lc, s);
    return hipErrorInvalidValue;
}
static int g_last_node_kernel = 0;    // family of the last node-MLP launch (tests / bench): 1 general, 2 16-row cooperative, 3 cooperative,
int last_node_kernel() { return g_last_node_kernel; }   // 5 k_node_split, 6 k_node_split<two sets>, 7 fp32-MFMA k_node_step, 8 / 9 16-row kernels on the split path, 10 k_node_split_h
hipError_t launch_node_step(int L, const NodeArgs& a, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    if (a.mode != 2) g_last_node_kernel = a.gen.use ? 1 : (a.c16 && L == 128 && a.chunk_t[0]) ? 2 : coop_ok(L, a.ntiles, a.chunk_t) ? 3 : 7;
    if (a.gen.use) {
        if (a.mode == 2) return launch_project(L, a, s);
        if (a.AGG2) {
            if (a.mode == 1) return hipErrorInvalidValue;      // two edge sets: the host projects per set
            DISPATCH_GEN(L, (k_node_step<4, 0, false, 2, true>), (k_node_step<2, 0, false, 2, true>), (k_node_step<1, 0, false, 2, true>), a, a.ntiles);
        }
        if (a.mode == 1) DISPATCH_GEN(L, (k_node_step<4, 0, true, 1, true>), (k_node_step<2, 0, true, 1, true>), (k_node_step<1, 0, true, 1, true>), a, a.ntiles);
        DISPATCH_GEN(L, (k_node_step<4, 0, false, 1, true>), (k_node_step<2, 0, false, 1, true>), (k_node_step<1, 0, false, 1, true>), a, a.ntiles);
    }
    if (a.c16 && L == 128 && a.chunk_t[0]) {
        LaunchCfg c16{2 * a.ntiles, 256, (size_t)4 * 8 * 64 * 16 + 2 * 64 * 4};
        if (g_fp32_split && (g_c16_split & 2) && a.split16[0] && (!a.AGG2 || a.split16[6])) {   // split path (pieces exchanged: 12 KiB per buffer)
            c16.lds = (size_t)4 * 12 * 64 * 16 + 2 * 64 * 4;
            if (a.mode != 2) g_last_node_kernel = 8;
            if (g_split_f16 && a.split16h[0] && (!a.AGG2 || a.split16h[6])) {      // two fp16 pieces
                if (a.bf) return a.AGG2 ? launch_k(k_node_coop16<2, true, 2>, a, c16, s) : launch_k(k_node_coop16<1, true, 2>, a, c16, s);
                return a.AGG2 ? launch_k(k_node_coop16<2, false, 2>, a, c16, s) : launch_k(k_node_coop16<1, false, 2>, a, c16, s);
            }
            if (a.bf) return a.AGG2 ? launch_k(k_node_coop16<2, true, 1>, a, c16, s) : launch_k(k_node_coop16<1, true, 1>, a, c16, s);
            return a.AGG2 ? launch_k(k_node_coop16<2, false, 1>, a, c16, s) : launch_k(k_node_coop16<1, false, 1>, a, c16, s);
        }
        if (a.bf) return a.AGG2 ? launch_k(k_node_coop16<2, true>, a, c16, s) : launch_k(k_node_coop16<1, true>, a, c16, s);
        return a.AGG2 ? launch_k(k_node_coop16<2, false>, a, c16, s) : launch_k(k_node_coop16<1, false>, a, c16, s);
    }
    const bool split_node = L == 128 && a.mode == 0 && a.split[0] && (!a.AGG2 || a.split[6]) && node_split_size(a.ntiles);
    // between the 16-row kernels' range and the split-path node kernels (8 k .. 16 k nodes on a mesh): the edge step ran a 32-row
    // kernel, the node side still fits half tiles of 16 rows two blocks per CU -- the 16-row node kernel on the split path, reading
    // the 32-edge-tile carry rows (16 k nodes: 37 -> 2x us against the cooperative 32-row kernel on the fp32 pipe)
    if (g_node16_mid && g_fp32_split && (g_c16_split & 2) && g_path == 0 && L == 128 && !a.bf && !a.AGG2 && a.mode != 2 && a.split16[0] &&
        coop_ok(L, a.ntiles, a.chunk_t) && !split_node) {
        LaunchCfg c16{2 * a.ntiles, 256, (size_t)4 * 12 * 64 * 16 + 2 * 64 * 4};
        g_last_node_kernel = 9;
        if (g_split_f16 && a.split16h[0]) return launch_k(k_node_coop16<1, false, 2, 5>, a, c16, s);
        return launch_k(k_node_coop16<1, false, 1, 5>, a, c16, s);
    }
    if (coop_ok(L, a.ntiles, a.chunk_t) && !split_node) {
        LaunchCfg c4{a.ntiles, 256, coop_lds()};
        if (coop_fence(a.ntiles)) return a.AGG2 ? launch_k(k_node_coop<true, true>, a, c4, s) : launch_k(k_node_coop<false, true>, a, c4, s);
        return a.AGG2 ? launch_k(k_node_coop<true, false>, a, c4, s) : launch_k(k_node_coop<false, false>, a, c4, s);
    }
    if (a.mode == 2) return launch_project(L, a, s);
    const bool proj = a.mode == 1;
    if (split_node) {   // split path (split.hip)
        LaunchCfg ls = tile_launch(L, a.ntiles, 2);
        ls.lds = (size_t)4 * 16384 * 2 + (size_t)T_COUNT * L * 4 + 64;
        if (g_split_f16 && a.splith[0] && !a.AGG2) {   // two fp16 pieces, three products
            g_last_node_kernel = 10;
            return launch_node_split_h(a, ls, s);
        }
        g_last_node_kernel = a.AGG2 ? 6 : 5;
        return launch_node_split(a, ls, s);
    }
    const int nres = resident_chunks(L, proj ? 6 : 4);
    const LaunchCfg lc = tile_launch(L, a.ntiles, nres);
    if (a.AGG2) {   // two edge sets: MLP only here, the host projects per set in separate launches
        if (proj) return hipErrorInvalidValue;
        if (L == 128) {
            if (small_launch(a.ntiles)) {
                LaunchCfg l0 = lc;
                l0.lds = (size_t)T_COUNT * L * 4 + 64;
                return launch_k(k_node_step<4, 0, false, 2>, a, l0, s);
            }
            return launch_k(k_node_step<4, 2, false, 2>, a, lc, s);
        }
        if (L == 64) return launch_k(k_node_step<2, 4, false, 2>, a, lc, s);
        if (L == 32) return launch_k(k_node_step<1, 4, false, 2>, a, lc, s);
        return hipErrorInvalidValue;
    }
    if (L == 128) {
        if (small_launch(a.ntiles)) {
            LaunchCfg l0 = lc;
            l0.lds = (size_t)T_COUNT * L * 4 + 64;
            return proj ? launch_k(k_node_step<4, 0, true>, a, l0, s) : launch_k(k_node_step<4, 0, false>, a, l0, s);
        }
        return proj ? launch_k(k_node_step<4, 2, true>, a, lc, s) : launch_k(k_node_step<4, 2, false>, a, lc, s);
    }
    if (L == 64) return proj ? launch_k(k_node_step<2, 6, true>, a, lc, s) : launch_k(k_node_step<2, 4, false>, a, lc, s);
    if (L == 32) return proj ? launch_k(k_node_step<1, 6, true>, a, lc, s) : launch_k(k_node_step<1, 4, false>, a, lc, s);
    return hipErrorInvalidValue;
}
// node MLP + residual + P / Q of the next step in one launch (split.hip: k_node_ring_hs) where launch_node_step + launch_project would run
// k_node_split_h + k_project_split_h: fp32, one edge set, two fp16 pieces, whole node range.  *launched = false: not this size / mode.
hipError_t launch_node_project_fused(int L, const NodeArgs& a, hipStream_t s, bool* launched) {
    *launched = false;
    if (!(node_ring_hs_enabled() && L == 128 && g_fp32_split == 1 && g_split_f16 && g_path == 0 && !a.bf && !a.AGG2 && !a.gen.use && a.mode == 0 &&
          a.tile0 == 0 && a.splith[0] && a.splith[4] && a.splith[5] && node_split_size(a.ntiles)))
        return hipSuccess;
    LaunchCfg ls = tile_launch(L, a.ntiles, 2);
    ls.threads = 512;
    int blocks = (a.ntiles + 7) / 8;
    if (blocks > num_cus()) blocks = num_cus();
    ls.blocks = ((blocks + NUM_XCD - 1) / NUM_XCD) * NUM_XCD;
    g_last_node_kernel = 11;
    *launched = true;
    return launch_node_ring_hs(a, ls, s);
}

hipError_t launch_project(int L, const NodeArgs& a, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    LaunchCfg lc = tile_launch(L, a.ntiles, 2);
    if (!a.gen.use && a.c16 && L == 128 && a.chunk_t[0] && !a.AGG2 && a.mode == 2) {
        LaunchCfg c16{2 * a.ntiles, 256, (size_t)4 * 8 * 64 * 16 + 2 * 64 * 4};
        if (g_fp32_split && (g_c16_split & 2) && a.split16[4]) {
            c16.lds = (size_t)4 * 12 * 64 * 16 + 2 * 64 * 4;
            if (g_split_f16 && a.split16h[4]) return a.bf ? launch_k(k_node_coop16<1, true, 2>, a, c16, s) : launch_k(k_node_coop16<1, false, 2>, a, c16, s);
            return a.bf ? launch_k(k_node_coop16<1, true, 1>, a, c16, s) : launch_k(k_node_coop16<1, false, 1>, a, c16, s);
        }
        return a.bf ? launch_k(k_node_coop16<1, true>, a, c16, s) : launch_k(k_node_coop16<1, false>, a, c16, s);
    }
    if (!a.gen.use && a.tile0 == 0 && a.mode == 2 && coop_ok(L, a.ntiles, a.chunk_t) && !(L == 128 && a.split[4] && node_split_size(a.ntiles))) {
        LaunchCfg c4{a.ntiles, 256, coop_lds()};
        if (coop_fence(a.ntiles)) return a.AGG2 ? launch_k(k_node_coop<true, true>, a, c4, s) : launch_k(k_node_coop<false, true>, a, c4, s);
        return a.AGG2 ? launch_k(k_node_coop<true, false>, a, c4, s) : launch_k(k_node_coop<false, false>, a, c4, s);
    }
    // (a.ntiles may be a part of the node tiles -- boundary / interior halves of a partition: the kernel family follows the part's size)
    const bool split_ok = !a.gen.use && a.mode == 2 && a.split[4] && node_split_size(a.ntiles);
    if (L == 128 && small_launch(a.ntiles) && !split_ok) {
        lc.lds = (size_t)T_COUNT * L * 4 + 64;
        return launch_k(k_project<4, false>, a, lc, s);
    }
    if (L == 128) {
        if (split_ok) {   // split path (split.hip)
            LaunchCfg ls = lc;
            ls.lds = (size_t)4 * 16384 * 2 + (size_t)T_COUNT * L * 4 + 64;
            if (g_split_f16 && a.splith[4]) return launch_project_split_h(a, ls, s);
            return launch_project_split(a, ls, s);
        }
        if (lc.threads == 512) lc.threads = MGN_PROJ_WAVES * 64;
        return launch_k(k_project<4, true>, a, lc, s);
    }
    if (L == 64) return launch_k(k_project<2, true>, a, lc, s);
    if (L == 32) return launch_k(k_project<1, true>, a, lc, s);
    return hipErrorInvalidValue;
}
hipError_t launch_enc_node(int L, const EncNodeArgs& a, hipStream_t s) {
    if (a.gen.use) DISPATCH_GEN(L, (k_enc_node<4, 0, true>), (k_enc_node<2, 0, true>), (k_enc_node<1, 0, true>), a, a.ntiles);
    if (a.ntiles > 0 && L == 128 && coop_size(a.ntiles, false)) {
        LaunchCfg c4{a.ntiles, 256, coop_lds()};
        if (g_fp32_split && g_split_f16 && a.splith[0]) return coop_fence(a.ntiles) ? launch_k(k_enc_node_coop<true, true>, a, c4, s) : launch_k(k_enc_node_coop<false, true>, a, c4, s);
        return coop_fence(a.ntiles) ? launch_k(k_enc_node_coop<true>, a, c4, s) : launch_k(k_enc_node_coop<false>, a, c4, s);
    }
    DISPATCH_L(k_enc_node, 4, a, a.ntiles);
}
hipError_t launch_enc_edge(int L, const EncEdgeArgs& a, hipStream_t s) {
    if (a.gen.use) DISPATCH_GEN(L, (k_enc_edge<4, 0, true>), (k_enc_edge<2, 0, true>), (k_enc_edge<1, 0, true>), a, a.ntiles);
    if (a.ntiles > 0 && L == 128 && coop_size(a.ntiles, true)) {
        LaunchCfg c4{a.ntiles, 256, coop_lds()};
        if (g_fp32_split && g_split_f16 && a.splith[0]) return coop_fence(a.ntiles) ? launch_k(k_enc_edge_coop<true, true>, a, c4, s) : launch_k(k_enc_edge_coop<false, true>, a, c4, s);
        return coop_fence(a.ntiles) ? launch_k(k_enc_edge_coop<true>, a, c4, s) : launch_k(k_enc_edge_coop<false>, a, c4, s);
    }
    DISPATCH_L(k_enc_edge, 2, a, a.ntiles);
}
hipError_t launch_decode(int L, const DecArgs& a, hipStream_t s) {
    if (a.gen.use) DISPATCH_GEN(L, (k_decode<4, 0, true>), (k_decode<2, 0, true>), (k_decode<1, 0, true>), a, a.ntiles);
    if (a.ntiles > 0 && L == 128 && coop_size(a.ntiles, false)) {
        LaunchCfg c4{a.ntiles, 256, coop_lds()};
        if (g_fp32_split && g_split_f16 && a.splith[0]) return coop_fence(a.ntiles) ? launch_k(k_decode_coop<true, true>, a, c4, s) : launch_k(k_decode_coop<false, true>, a, c4, s);
        return coop_fence(a.ntiles) ? launch_k(k_decode_coop<true>, a, c4, s) : launch_k(k_decode_coop<false>, a, c4, s);
    }
    DISPATCH_L(k_decode, 2, a, a.ntiles);
}

static LaunchCfg bf_launch(int ntiles, int nchunks) {
    LaunchCfg lc = tile_launch(128, ntiles, 0);
    lc.lds = (size_t)nchunks * BF_CH * 2 + (size_t)T_COUNT * 128 * 4;
    return lc;
}
static int g_bf_edge = [] { const char* e = getenv("MGN_BF_EDGE"); return e ? atoi(e) : 1; }();   // 0: first-generation kernel (A/B)
hipError_t launch_edge_bf16(const BfEdgeArgs& a, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    LaunchCfg lc = bf_launch(a.ntiles, 3);
    if (g_bf_edge) return launch_k(k_edge_bf16_pipe, a, lc, s);   // software-pipelined: two waves per SIMD, 256 registers
    if (lc.threads == 512) lc.threads = MGN_BF_WAVES * 64;   // large launch: more waves per SIMD hide the memory phases
    return launch_k(k_edge_bf16, a, lc, s);
}
static int g_bf_node = [] { const char* e = getenv("MGN_BF_NODE"); return e ? atoi(e) : 1; }();   // 0: first-generation node kernels (A/B)
hipError_t launch_node_bf16(const BfNodeArgs& a, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    return g_bf_node ? launch_k(k_node_bf16_pipe, a, bf_launch(a.ntiles, 4), s) : launch_k(k_node_bf16, a, bf_launch(a.ntiles, 4), s);
}
hipError_t launch_project_bf16(const BfNodeArgs& a, hipStream_t s) {
    if (a.ntiles <= 0) return hipSuccess;
    return g_bf_node ? launch_k(k_project_bf16_pipe, a, bf_launch(a.ntiles, 2), s) : launch_k(k_project_bf16, a, bf_launch(a.ntiles, 2), s);
}
hipError_t launch_tile_f32_to_bf16(const float* src, uint16_t* dst, int64_t ntiles, hipStream_t s) {
    const int64_t n = ntiles * 512;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tile_f32_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}
hipError_t launch_tile_bf16_to_f32(const uint16_t* src, float* dst, int64_t ntiles, hipStream_t s) {
    const int64_t n = ntiles * 512;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tile_bf16_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}
hipError_t launch_gather_rows16(const uint16_t* src, const int32_t* idx, uint16_t* dst, int64_t rows, int dst_stride, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int64_t n = rows * 16;
    hipLaunchKernelGGL(k_gather_rows16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, idx, dst, rows, dst_stride / 8);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const float* src, const int32_t* idx, float* dst, int64_t rows, int L, int dst_stride, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int64_t n = rows * (L / 4);
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, idx, dst, rows, L / 4, dst_stride / 4);
    return hipGetLastError();
}

hipError_t launch_scatter_prows(const float* src, int src_stride, float* dst, int64_t row0, int64_t rows, int L, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int64_t n = rows * (L / 4);
    hipLaunchKernelGGL(k_scatter_prows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, src_stride / 4, dst, row0, rows, L / 4);
    return hipGetLastError();
}
hipError_t launch_scatter_prows16(const uint16_t* src, int src_stride, uint16_t* dst, int64_t row0, int64_t rows, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int64_t n = rows * 16;
    hipLaunchKernelGGL(k_scatter_prows16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, src_stride / 8, dst, row0, rows);
    return hipGetLastError();
}
bool prows_blocked() { return MGN_PROW_BLOCK != 0; }
hipError_t launch_rows_to_tiles(const float* src, const int64_t* gid64, const int32_t* gid32, float* dst, int64_t rows, int L, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int64_t n = rows * (L / 4);
    hipLaunchKernelGGL(k_rows_to_tiles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, gid64, gid32, dst, rows, L);
    return hipGetLastError();
}
hipError_t launch_tiles_to_rows(const float* src, const int64_t* gid64, const int32_t* gid32, float* dst, int64_t rows, int L, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int64_t n = rows * (L / 4);
    hipLaunchKernelGGL(k_tiles_to_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, gid64, gid32, dst, rows, L);
    return hipGetLastError();
}

hipError_t launch_randn_rows(float* dst, const int64_t* gid64, const int32_t* gid32, int64_t rows, int L, uint64_t seed,
                             hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const int64_t n = ((rows + TILE - 1) / TILE) * TILE * L;
    hipLaunchKernelGGL(k_randn_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, gid64, gid32, rows, L, seed);
    return hipGetLastError();
}

int checksum_partials() { return 2 * CHECKSUM_BLOCKS; }

hipError_t launch_checksum(const float* src, int64_t n, double* partials, hipStream_t s) {
    hipLaunchKernelGGL(k_checksum, dim3(CHECKSUM_BLOCKS), dim3(256), 0, s, src, n < 0 ? 0 : n, partials);
    return hipGetLastError();
}

int kernels_prow_block() { return MGN_PROW_BLOCK; }

}  // namespace mgn
