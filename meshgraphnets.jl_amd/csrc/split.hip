// fp32 processor kernels on the bf16 matrix cores at fp32 accuracy ("exact split" path).
//
// Replaces, for large fp32 launches at L = 128, the three fp32-MFMA kernels of a processor step (kernels.hip: k_edge_step,
// k_node_step, k_project -- reference: the Processor of GraphNetCore called at src/solve.jl:200).  Storage, tables, LayerNorm,
// residual, segmented scan, carry rows, launch geometry are those kernels'; only the L x L layers differ:
//   * every fp32 operand is the exact sum of three bf16 values (x = x1 + x2 + x3, 3 x 8 significand bits); of the nine piece
//     products of a * b the six with i + j <= 4 carry everything above 2^-24 relative; each product of two bf16 values is exact
//     in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32: one L x L layer against float64 has max relative error 1.7e-7
//     (a plain fp32 GEMM: 4.2e-7).  Six bf16 MFMAs cost 6 / 16 of the matrix time of the fp32 MFMA.
//   * weights are split on the host (mgn_set_params: pack_chunk_split, [piece][s][t][lane][8 bf16]); activations are split IN the
//     MFMA stream: the pieces of k-step s + 1 are computed (11 VALU instructions per pair of values, ReLU folded in) between the
//     24 MFMAs of k-step s, where they issue in the shadow of the matrix pipe (tools/overlap_probe.hip: four VALU instructions
//     per v_mfma_f32_32x32x16_bf16 are free, a separate split phase is not).  Only two k-steps' pieces are live, which leaves
//     registers for rings of 6-8 streamed weight fragments per piece (an L2 round trip under load is ~1.5 k cycles).
//   * the hi pieces of all chunks (+ one mid piece) are LDS-resident, the rest stream from L2 through those rings.
// The accumulator layout of the two MFMA shapes is the same; element j of lane half h at k-step s is accumulator register
// 8 (s & 1) + j of block s >> 1 (the bf16 kernels' correspondence), so a layer's output feeds the next layer without a shuffle.
// (Packed fp32 VALU holds the matrix pipe for ~12 cycles per instruction, tools/overlap_probe.hip: none may appear inside a chain.  The
// split's subtractions sit between inline-asm conversions and scheduling fences, where hipcc's SLP vectoriser does not pair them --
// the shipped object has its v_pk_* instructions in the epilogue only -- so no -fno-slp-vectorize is passed; MGN_SPLIT_FLAGS adds it
// for A/B runs.)
#include <cstdlib>

#include "kernels.h"
#include "tile_common.hpp"
#include "split_common.hpp"

namespace mgn {

#ifndef MGN_SP_CARRY
#define MGN_SP_CARRY 1            // k_node_split, k_project_split: weight rings carried from chain to chain (needs MGN_SP_BUFFER)
#endif
#ifndef MGN_SP_BUFFER
#define MGN_SP_BUFFER 1           // sp_layer_otf: streamed weight pieces through buffer descriptors (0: 64-bit pointers)
#endif

// One L x L layer: acc += W^T in, `in` split on the fly.  p1 / p2 / p3: the chunk's hi / mid / lo piece ([s][t][lane] fragments of
// 8 bf16).  p1 is LDS-resident; G2 / G3: p2 / p3 stream from L2 through register rings D (s, t) groups deep (else LDS too).  The
// rings are pinned by scheduling fences: left to itself hipcc requests every streamed fragment one MFMA before its use.
// The rings can be carried from chain to chain (SpRing, CARRY = 1): fragment f of a chain lives in slot (f + OFF) % D, and while the
// last D steps of this chain release their slots the FIRST D fragments of the next chain (nx2 / nx3: its mid / lo pieces) are requested
// into them -- its ring is full when it starts (offset (OFF + 32) % D) instead of every chain opening with an L2 round trip.
// PRIMED: the ring came that way; NEXT: this chain primes the one behind it.
template <int D> struct SpRing { u32x4 r2[D], r3[D]; };
template <bool G2, bool G3, bool RELU, int D, bool G1 = false, int OFF = 0, bool PRIMED = false, bool NEXT = false>
DEVINL void sp_layer_otf(f32x16 (&acc)[4], const f32x16 (&in)[4], const u32x4* p1, const u32x4* p2, const u32x4* p3, int lane,
                         SpRing<D>* carry = nullptr, const u32x4* nx2 = nullptr, const u32x4* nx3 = nullptr) {
    const u32x4* w1 = p1 + lane;
    const u32x4* w2 = p2 + lane;
    const u32x4* w3 = p3 + lane;
#if MGN_SP_BUFFER
    // streamed pieces through buffer descriptors: the lane's offset is ONE register for all of them, the fragment index a scalar
    const N16Buf b1 = n16_buf(G1 ? p1 : nullptr), b2 = n16_buf(G2 ? p2 : nullptr), b3 = n16_buf(G3 ? p3 : nullptr);
    const unsigned voff = (unsigned)lane * 16u;
#define SP_G1(IDX) n16_ldu(b1, voff, (IDX) * 1024)
#define SP_G2(IDX) n16_ldu(b2, voff, (IDX) * 1024)
#define SP_G3(IDX) n16_ldu(b3, voff, (IDX) * 1024)
#else
#define SP_G1(IDX) w1[(IDX) * 64]
#define SP_G2(IDX) w2[(IDX) * 64]
#define SP_G3(IDX) w3[(IDX) * 64]
#endif
    static_assert(!(PRIMED || NEXT) || (MGN_SP_BUFFER && !G1), "ring carry-over is written for the descriptor path, mid / lo streams");
    u32x4 r1[G1 ? D : 1], r2[G2 ? D : 1], r3[G3 ? D : 1];      // G1: the hi piece streams from L2 as well (a chunk that has no room in LDS)
    if constexpr (PRIMED) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if constexpr (G2) r2[d] = carry->r2[d];
            if constexpr (G3) r3[d] = carry->r3[d];
        }
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if constexpr (G1) r1[d] = SP_G1(d);
            if constexpr (G2) r2[(d + OFF) % D] = SP_G2(d);
            if constexpr (G3) r3[(d + OFF) % D] = SP_G3(d);
        }
    }
#if MGN_SP_BUFFER
    const N16Buf c2 = n16_buf(NEXT && G2 ? nx2 : nullptr), c3 = n16_buf(NEXT && G3 ? nx3 : nullptr);
#endif
    u32x4 n1, n2, n3;
    if constexpr (!G1) n1 = w1[0];
    if constexpr (!G2) n2 = w2[0];
    if constexpr (!G3) n3 = w3[0];
    SpPieces p;                                         // the pieces of k-step s
#pragma unroll
    for (int u = 0; u < 4; ++u) sp_split_pair<RELU>(p.h[u], p.m[u], p.l[u], in[0][2 * u], in[0][2 * u + 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        SpPieces n;                                     // ... and of k-step s + 1, one pair per (s, t) group
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int it = 4 * s + t;
            u32x4 a1, a2, a3;
            if constexpr (G1) a1 = r1[it % D]; else a1 = n1;
            if constexpr (G2) a2 = r2[(it + OFF) % D]; else a2 = n2;
            if constexpr (G3) a3 = r3[(it + OFF) % D]; else a3 = n3;
            if (it + 1 < 32) {
                if constexpr (!G1) n1 = w1[(it + 1) * 64];
                if constexpr (!G2) n2 = w2[(it + 1) * 64];
                if constexpr (!G3) n3 = w3[(it + 1) * 64];
            }
            if (it + D < 32) {
                if constexpr (G1) r1[it % D] = SP_G1(it + D);
                if constexpr (G2) r2[(it + OFF) % D] = SP_G2(it + D);
                if constexpr (G3) r3[(it + OFF) % D] = SP_G3(it + D);
            } else if constexpr (NEXT) {
#if MGN_SP_BUFFER
                if constexpr (G2) r2[(it + OFF) % D] = n16_ldu(c2, voff, (it + D - 32) * 1024);
                if constexpr (G3) r3[(it + OFF) % D] = n16_ldu(c3, voff, (it + D - 32) * 1024);
#endif
            }
            if (s < 7) {
                const int sn = s + 1;
                sp_split_pair<RELU>(n.h[t], n.m[t], n.l[t], in[sn >> 1][8 * (sn & 1) + 2 * t], in[sn >> 1][8 * (sn & 1) + 2 * t + 1]);
            }
            const sp_bf16x8 bh = sp_op(p.h), bm = sp_op(p.m), bl = sp_op(p.l);
#ifdef MGN_WHATIF_MFMA16_NODE   // diagnostic (wrong results): matrix time and operand traffic of this layer on v_mfma_f32_16x16x32_bf16
#define OTF_MFMA(A_, B_)                                                                                               \
            do {                                                                                                       \
                f32x4 c0_, c1_;                                                                                        \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { c0_[i_] = acc[t][i_]; c1_[i_] = acc[t][4 + i_]; }     \
                c0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, c0_, 0, 0, 0);                                    \
                c1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, c1_, 0, 0, 0);                                    \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { acc[t][i_] = c0_[i_]; acc[t][4 + i_] = c1_[i_]; }     \
            } while (0)
#else
#define OTF_MFMA(A_, B_) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, acc[t], 0, 0, 0)
#endif
            OTF_MFMA(sp_wop(a3), bh);      // small terms first
            OTF_MFMA(sp_wop(a2), bm);
            OTF_MFMA(sp_wop(a1), bl);
            OTF_MFMA(sp_wop(a2), bh);
            OTF_MFMA(sp_wop(a1), bm);
            OTF_MFMA(sp_wop(a1), bh);
#undef OTF_MFMA
#if MGN_SP2_INTERLEAVE
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        p = n;
    }
    if constexpr (NEXT) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if constexpr (G2) carry->r2[d] = r2[d];
            if constexpr (G3) carry->r3[d] = r3[d];
        }
    }
}

#undef SP_G1
#undef SP_G2
#undef SP_G3

#ifndef MGN_SP2_D
#define MGN_SP2_D 6
#endif
#ifndef MGN_NODE_VNEXT_FIRST
#define MGN_NODE_VNEXT_FIRST 0   // node-side split kernels: 1 = the next tile's V requested ahead of this tile's stores (1.178 vs 1.170 ms: nothing; the waves of these kernels are not in lock-step)
#endif
#ifndef MGN_SP2_D1
#define MGN_SP2_D1 8       // layer 2 streams one piece only
#endif
// ================================================================================================
// Lock-step variant of the edge step: the streamed weight pieces go through ONE LDS ring per block instead of eight per-wave
// register rings.  The register-ring kernel (k_edge_split2, docs/experiments/split_variants_r04.hip) reads 160 KiB of weight pieces per tile and wave from L2; stamps show what that costs: with ~100
// KiB of requests queued per CU every other load of the tile (gathered P / Q rows, the e tile) waits thousands of cycles behind
// them (38 k of a tile's 76 k cycles are its epilogue).  Here the eight waves of a block walk their tiles in lock-step, so that a
// fragment fetched once serves all eight:
//   * resident in LDS: the hi pieces of W1e, W2, W3 (96 KiB) + tables;
//   * the mid and lo pieces of the layer in progress stream through three 16 KiB window buffers (8 (s, t) steps of both pieces):
//     during window w every thread requests its 2 x 16 bytes of window w + 2 at the window's first step and stores them at its
//     seventh; a barrier closes every window (12 per tile).  Three buffers, so that the one-step-ahead fragment reads may cross a
//     window boundary: what they touch was written a whole window earlier.  12 windows per tile = 0 mod 3: the buffer of a window
//     is the same for every tile.
// L2 weight traffic per tile: 192 KiB per EIGHT tiles.  All waves run the same number of tiles (stores of padding tiles masked).
//
// Turnover (round 4, MGN_RING_ENEXT = 1).  s_waitcnt vmcnt retires in order and counts stores: a load requested behind the tile's 16 KiB
// of e stores is not "back" before those stores are acknowledged, so round 3's order -- residual, store e, THEN request the next tile's e
// and Q rows -- put the store tail (6-15 k cycles for sixteen store instructions of eight lock-step waves) in front of the next tile's
// first MFMA.  Now layer 3's refill brings the NEXT tile's e tile into the registers its input releases (twelve pieces inside the layer,
// the last four at the start of the epilogue: requested inside the layer they are spilled where they land, with an s_waitcnt vmcnt(0) in
// the chain), this tile's e is read a second time at the start of the epilogue into the registers the chain has just released (pieces,
// fragments, loader staging: the Infinity Cache serves it while the LayerNorm runs), and only the Q gather is left behind the stores.
// No scratch access in the tile loop (round 3: 16 spilled registers, three `s_waitcnt vmcnt(0)` + scratch_store in layer 2's chain).
// M-1M, same box: 3.42-3.45 -> 3.11-3.13 ms.  What-if builds on this order (wrong results): no e store 2.80, e stores into one cached
// tile per wave 3.02 (so 0.27 of the 0.36 ms the stores cost is their issue, not HBM), no aggregate stores 3.03, aggregate rows stored
// row-major 3.13, every stream cached 2.60.
// ================================================================================================
// (Tried and dropped, docs/experiments.md: two groups of four waves half a tile apart, with slot barriers or with group-local
// LDS-counter barriers: 4.0-4.1 ms against 3.4 -- the epilogue's memory round trips then pace the other group's chains.)
template <int W>
struct Rg {
    static constexpr int WPL = 32 / W;          // windows per layer
    static constexpr int NW = 3 * WPL;          // windows per tile (a multiple of 3: window -> buffer is the same for every tile)
    static constexpr int BUF = 2 * W * 64;      // u32x4 elements per window buffer: [piece (mid, lo)][step][lane]
};
struct RingSrc {
    const u32x4* mid[3];                        // global mid / lo pieces of layers 1..3 (W1e, W2, W3)
    const u32x4* lo[3];
};
struct RingFrag {
    u32x4 h, m, l;                              // fragments of the next step (read one step ahead)
};
DEVINL void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// first fragments of a layer (hi from its resident piece, mid / lo from the ring window WPL * LYR); `ring` carries the lane offset
template <int W, int LYR>
DEVINL RingFrag ring_first(const u32x4* hi, const u32x4* ring, int lane) {
    constexpr int b = (Rg<W>::WPL * LYR) % 3;
    RingFrag f;
    f.h = hi[lane];
    f.m = ring[b * Rg<W>::BUF];
    f.l = ring[b * Rg<W>::BUF + W * 64];
    return f;
}
// One L x L layer (layer LYR of the tile).  nx: the fragments of step 0 in, those of the next layer's step 0 out (LYR < 2;
// hi_next = that layer's resident hi piece).  tid: thread index within the group that shares the ring.
// Refill (RFS > 0): `in` is consumed k-step by k-step, and what a tile needs next from memory -- the P rows that complete layer 1,
// the e tile for the residual behind layer 3 -- is 64 registers wide, which nothing has to spare.  So the registers of k-step s are
// reloaded, as soon as the split has consumed them, with the two 16-byte pieces of k-step s + 2 of rf (piece m at rf[m * RFS]);
// the pieces of k-steps 0 and 1 wait in a 16-register side buffer.  The last request goes out two k-steps before the layer ends;
// `in` comes back holding rf's row in fragment order.  (In the epilogue these loads were two exposed memory round trips.)
// When the requests go out (MGN_RING_REFILL_AT_REQUEST): 0 = two behind every k-step, side buffer at the top of the layer (the first
// version); 1 = in batches of four right behind the window requests (vmcnt retires in order, so a request there is first waited for
// at the NEXT window's LDS store, 5.4 k cycles later instead of 2.3-4.2 k): no gain, the e tile's latency is not what costs;
// 2 (default) = one request every second step, and layer 3's side buffer requested the same way inside layer 2's last two k-steps
// (NRFS, rf_next): 3.53 -> 3.46 ms.  What a request costs in the chain is its ISSUE: the eight waves are in lock-step, their
// requests reach the CU's one memory pipeline together, and a wave whose request is not accepted issues no MFMA either (stamps:
// every request adds ~500 cycles to its layer whether it hits L2 or not).
// QP (layer 1, MGN_RING_QPSTREAM): the gathered rows P[s] and Q[r] that complete the layer stream THROUGH it instead of waiting in 64 + 64
// registers: the accumulator starts from zero, every 16-byte piece is requested eight (s, t) steps before it is added, and it is added
// into accumulator t + 2 while the MFMAs of step (s, t) run on accumulator t (written two steps ago, read again in two).  Per
// accumulator: its four P pieces are added at k-steps 2 .. 5, its four Q pieces two by two at k-steps 6 and 7.  Nothing of the
// turnover is then requested behind the tile's e stores, and no register waits for a row.
template <int W, int LYR, bool RELU, int RFS = 0, int NRFS = 0, int NWV = 8, bool WRAP = false, bool QP = false>
DEVINL void sp_layer_ring(f32x16 (&acc)[4], f32x16 (&in)[4], const u32x4* hi, const u32x4* hi_next, u32x4* ring, const RingSrc& src,
                          RingFrag& nx, int lane, int tid, const f32x4* rf = nullptr, f32x4* side = nullptr,
                          const f32x4* rf_next = nullptr, const f32x4* qp_p = nullptr, const f32x4* qp_q = nullptr) {
#ifndef MGN_RING_ROT
#define MGN_RING_ROT 2
#endif
#ifndef MGN_RING_REFILL_AT_REQUEST
#define MGN_RING_REFILL_AT_REQUEST 2
#endif
    constexpr int ROT = MGN_RING_ROT;            // k-steps between a register's release and the use of what it is refilled with
#ifndef MGN_RING_SIDE_EARLY
#define MGN_RING_SIDE_EARLY 1                // bit 0: layer 3's side buffer requested inside layer 2, bit 1: layer 1's in the epilogue before
#endif
    constexpr int MODE = MGN_RING_REFILL_AT_REQUEST;                  // 0: two requests behind each k-step, 1: with the window requests, 2: one per two steps
    constexpr bool ATREQ = MODE == 1;
    static_assert(MODE == 0 || ROT == 2, "the other schedules are written for a rotation of two k-steps");
    // WRAP (schedule 2 only; for a refill that is first used well after the layer, the e tile behind layer 3): no side buffer and no
    // rotation -- the registers of k-step s take the pieces of k-step s, for s < 8 - ROT; the last ROT k-steps' pieces (in[3][16 - 8 ROT ..])
    // are left to the caller, who requests them behind the layer (requested inside it they are spilled where they land, with an
    // s_waitcnt vmcnt(0) in the middle of the chain: the 64 registers of `in` are free only when the layer is done).
    static_assert(!WRAP || (MODE == 2 && RFS > 0), "the plain refill is written for schedule 2");
    f32x4 side_local[2 * ROT];
    if constexpr (!WRAP && RFS > 0 && (MODE == 0 || !((MGN_RING_SIDE_EARLY >> (LYR == 0 ? 1 : 0)) & 1))) {
        side = side_local;
#pragma unroll
        for (int m = 0; m < 2 * ROT; ++m) side[m] = rf[m * RFS];
    }
    constexpr int WPL = Rg<W>::WPL, NW = Rg<W>::NW, BUF = Rg<W>::BUF;
    SpPieces p;
#pragma unroll
    for (int u = 0; u < 4; ++u) sp_split_pair<RELU>(p.h[u], p.m[u], p.l[u], in[0][2 * u], in[0][2 * u + 1]);
    constexpr int LPT = 8 / NWV;                 // fragments per piece and thread in a window (NWV waves share the loading)
    u32x4 ld_m[LPT], ld_l[LPT];                  // this thread's share of window gw + 2 on its way to LDS
    unsigned voff = (unsigned)tid * 16u;
    asm volatile("" : "+v"(voff));
    f32x4 qp[QP ? 4 : 1][QP ? 8 : 1];            // QP: the pieces in flight, per accumulator (SSA values: live from request to add)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        SpPieces n;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int it = 4 * s + t;
            const int gw = WPL * LYR + it / W;                        // global window of this step
            const u32x4 a1 = nx.h, a2 = nx.m, a3 = nx.l;
            if constexpr (QP) {
                constexpr int STR = STRIDE_PROW;
                const int tt = (t + 2) & 3;                            // the accumulator these pieces belong to
                if (s < 4) {
                    qp[tt][s] = qp_p[(4 * tt + s) * STR];
                } else if (s < 6) {
                    qp[tt][4 + 2 * (s - 4)] = qp_q[(4 * tt + 2 * (s - 4)) * STR];
                    qp[tt][5 + 2 * (s - 4)] = qp_q[(4 * tt + 2 * (s - 4) + 1) * STR];
                }
                if (s >= 2 && s < 6) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[tt][4 * (s - 2) + i] += qp[tt][s - 2][i];
                } else if (s >= 6) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[tt][4 * (2 * (s - 6) + u) + i] += qp[tt][4 + 2 * (s - 6) + u][i];
                }
            }
            if (it % W == 0) {                                         // request window gw + 2
                const int g2 = (gw + 2) % NW, l2 = g2 / WPL, w2 = g2 % WPL;
                // (uniform base + 32-bit lane offset: the scalar-base form of global_load; as per-thread 64-bit pointers
                // hipcc hoists them out of the tile loop and spills them)
#if !(defined(MGN_WHATIF_LOADER) && (MGN_WHATIF_LOADER & 1))     // diagnostic (wrong results): 1 = no window loads at all, 2 = no barriers
#pragma unroll
                for (int i = 0; i < LPT; ++i) {
                    ld_m[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src.mid[l2] + w2 * W * 64 + i * NWV * 64) + voff);
                    ld_l[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src.lo[l2] + w2 * W * 64 + i * NWV * 64) + voff);
                }
#endif
                if constexpr (ATREQ && W == 8) {
                    if constexpr (RFS > 0) {
                        // the registers of k-steps s - 2 and s - 1 (released one and two k-steps ago: refilling k-step s's own as
                        // well leaves the allocator nothing to work with, 70 spills)
                        const int lo_k = s - 2, hi_k = s - 1;
#pragma unroll
                        for (int k = 0; k < 6; ++k)
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                if (k < lo_k || k > hi_k) continue;
                                const f32x4 v = rf[(2 * (k + ROT) + u) * RFS];
#pragma unroll
                                for (int i = 0; i < 4; ++i) in[k >> 1][8 * (k & 1) + 4 * u + i] = v[i];
                            }
                    }
                    if constexpr (NRFS > 0 && (MGN_RING_SIDE_EARLY & 1)) {
                        if (s == 6) {
#pragma unroll
                            for (int m = 0; m < 2 * ROT; ++m) side[m] = rf_next[m * NRFS];
                        }
                    }
                }
            }
            if constexpr (MODE == 2) {
                if constexpr (RFS > 0) {
                    if ((t & 1) && s < 8 - ROT) {                      // registers of k-step s (free since the step began), half t >> 1
                        const f32x4 v = rf[(2 * (WRAP ? s : s + ROT) + (t >> 1)) * RFS];
#pragma unroll
                        for (int i = 0; i < 4; ++i) in[s >> 1][8 * (s & 1) + 4 * (t >> 1) + i] = v[i];
                    }
                }
                if constexpr (NRFS > 0 && (MGN_RING_SIDE_EARLY & 1)) {
                    if ((t & 1) && s >= 6) side[2 * (s - 6) + (t >> 1)] = rf_next[(2 * (s - 6) + (t >> 1)) * NRFS];
                }
            }
            if (it + 1 < 32) {
                const int gn = WPL * LYR + (it + 1) / W;
                nx.h = hi[(it + 1) * 64 + lane];
                nx.m = ring[(gn % 3) * BUF + ((it + 1) % W) * 64];
                nx.l = ring[(gn % 3) * BUF + W * 64 + ((it + 1) % W) * 64];
            } else if (LYR < 2) {
                nx = ring_first<W, (LYR + 1) % 3>(hi_next, ring, lane);   // (that window was written two windows ago)
            }
            if (it % W == W - 2) {                                     // ... and store it: its buffer was last read in window gw - 1
                const int b2 = (gw + 2) % 3;
#if !(defined(MGN_WHATIF_LOADER) && (MGN_WHATIF_LOADER & 1))
#pragma unroll
                for (int i = 0; i < LPT; ++i) {
                    ring[b2 * BUF + i * NWV * 64 + tid - lane] = ld_m[i];
                    ring[b2 * BUF + W * 64 + i * NWV * 64 + tid - lane] = ld_l[i];
                }
#endif
            }
            if (s < 7) {
                const int sn = s + 1;
                sp_split_pair<RELU>(n.h[t], n.m[t], n.l[t], in[sn >> 1][8 * (sn & 1) + 2 * t], in[sn >> 1][8 * (sn & 1) + 2 * t + 1]);
            }
            const sp_bf16x8 bh = sp_op(p.h), bm = sp_op(p.m), bl = sp_op(p.l);
#ifdef MGN_WHATIF_MFMA16    // diagnostic (wrong results): the same operand traffic and matrix time on v_mfma_f32_16x16x32_bf16 -- what the
                            // other shape's clock is worth to this kernel before anyone re-writes its layouts
#define RING_MFMA(A_, B_)                                                                                          \
            do {                                                                                                   \
                f32x4 c0_, c1_;                                                                                    \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { c0_[i_] = acc[t][i_]; c1_[i_] = acc[t][4 + i_]; } \
                c0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, c0_, 0, 0, 0);                                \
                c1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, c1_, 0, 0, 0);                                \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { acc[t][i_] = c0_[i_]; acc[t][4 + i_] = c1_[i_]; } \
            } while (0)
#else
#define RING_MFMA(A_, B_) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, acc[t], 0, 0, 0)
#endif
            RING_MFMA(sp_wop(a3), bh);      // small terms first
            RING_MFMA(sp_wop(a2), bm);
            RING_MFMA(sp_wop(a1), bl);
            RING_MFMA(sp_wop(a2), bh);
            RING_MFMA(sp_wop(a1), bm);
            RING_MFMA(sp_wop(a1), bh);
#undef RING_MFMA
            __builtin_amdgcn_sched_barrier(0);
#if !(defined(MGN_WHATIF_LOADER) && (MGN_WHATIF_LOADER & 2))
            if (it % W == W - 1) ring_barrier();                       // window closed: every wave has read it, window gw + 2 is in LDS
#endif
        }
        p = n;
        if constexpr (RFS > 0 && MODE == 0) {
            // (the eight waves of a block run in lock-step: without a per-wave delay all sixteen gather instructions of a k-step
            // reach the CU's memory pipeline at once; 64 line visits each)
            if (s < 8 - ROT) {                                         // registers of k-step s <- pieces of k-step s + ROT
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4 v = rf[(2 * (s + ROT) + u) * RFS];
#pragma unroll
                    for (int i = 0; i < 4; ++i) in[s >> 1][8 * (s & 1) + 4 * u + i] = v[i];
                }
            }
        }
    }
    if constexpr (RFS > 0 && !WRAP) {                                  // un-rotate: k-step u's pieces sit in the registers of k-step u - 2
        f32x16 r[4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r[u >> 1][8 * (u & 1) + j] = u < ROT ? side[2 * u + (j >> 2)][j & 3] : in[(u - ROT) >> 1][8 * ((u - ROT) & 1) + j];
#pragma unroll
        for (int t = 0; t < 4; ++t) in[t] = r[t];
    }
}

// one level of the segmented scan (tile_common.hpp: segmented_scan) on the whole fragment
#define RG_SCAN_LEVEL(ACC, COND, CTRL)                                                                                       \
    do {                                                                                                                     \
        const float m_ = (COND) ? 1.f : 0.f;                                                                                 \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)                                                                     \
            _Pragma("unroll") for (int k_ = 0; k_ < 16; ++k_)                                                                \
                asm volatile("v_fmac_f32_dpp %0, %0, %1 " CTRL " bound_ctrl:0" : "+v"(ACC[t_][k_]) : "v"(m_));               \
    } while (0)

// the e tile leaves the chip once per step and comes back 3 GB later: cache policy of its stores / loads (A/B switches)
#ifndef MGN_RING_ESTORE
#define MGN_RING_ESTORE 0        // 0 plain, 1 nt, 2 sc1, 3 sc0 sc1 (2, 3: inline assembly -- A/B ONLY: the compiler does not see a store there, so neither its
#endif                           // hazard recogniser nor its counters do; with 3 the eight-partition M-1M test lost reproducibility.  The product's write-through stores: MGN_RING_EST_BUF)
#ifndef MGN_RING_ELOAD
#define MGN_RING_ELOAD 0         // 0 plain, 1 nt
#endif
DEVINL void ring_store_e(f32x4* p, const f32x16 (&x)[4]) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = x[m >> 2][4 * (m & 3) + i];
        f32x4* q = p + m * STRIDE_TILE;
        if constexpr (MGN_RING_ESTORE == 1) __builtin_nontemporal_store(v, q);
        else if constexpr (MGN_RING_ESTORE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
        else if constexpr (MGN_RING_ESTORE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
        else *q = v;
    }
}
DEVINL void ring_store_e_piece(f32x4* q, f32x4 v) {
    if constexpr (MGN_RING_ESTORE == 1) __builtin_nontemporal_store(v, q);
    else if constexpr (MGN_RING_ESTORE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
    else if constexpr (MGN_RING_ESTORE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
    else *q = v;
}
// store_frag with a cache policy: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1 (write-through at system scope: the line does not stay dirty in L2)
template <int NT, int POL>
DEVINL void store_frag_pol(f32x4* __restrict__ p, int stride, const f32x16 (&x)[NT]) {
#pragma unroll
    for (int m = 0; m < 4 * NT; ++m) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = x[m >> 2][4 * (m & 3) + i];
        f32x4* q = p + m * stride;
        if constexpr (POL == 1) __builtin_nontemporal_store(v, q);
        else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
        else if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
        else *q = v;
    }
}
#ifndef MGN_AGG_STORE
#define MGN_AGG_STORE 0          // k_edge_ring_h: policy of the aggregate / carry-row stores
#endif
#ifndef MGN_NODE_STORE
#define MGN_NODE_STORE 0         // k_node_split_h: policy of the v stores
#endif
#ifndef MGN_PROJ_STORE
#define MGN_PROJ_STORE 0         // k_project_split_h: policy of the P / Q stores
#endif
DEVINL void ring_load_e(f32x16 (&x)[4], const f32x4* p) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const f32x4 v = MGN_RING_ELOAD ? __builtin_nontemporal_load(p + m * STRIDE_TILE) : p[m * STRIDE_TILE];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[m >> 2][4 * (m & 3) + i] = v[i];
    }
}
// pieces 12 .. 15 (k-steps 6, 7) of an e tile: what sp_layer_ring's plain refill (WRAP) leaves to its caller
DEVINL void ring_load_e_tail(f32x16 (&x)[4], const f32x4* p) {
#pragma unroll
    for (int m = 12; m < 16; ++m) {
        const f32x4 v = p[m * STRIDE_TILE];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[3][4 * (m & 3) + i] = v[i];
    }
}
#ifdef MGN_RING_EPI_STAMPS     // diagnostic: the eight stamp slots on the epilogue (1: chains done ... 7: turnover requested)
#define CST(k) do {} while (0)
#define EST(k) STAMP(k)
#else
#define CST(k) STAMP(k)
#define EST(k) do {} while (0)
#endif
// NWV waves per block: 8 (two per SIMD), or 4 (one per SIMD) for launches of a few rounds -- a round of 4-wave blocks takes half the
// tiles and less than half the time of one of 8-wave blocks (nothing shares its SIMD's matrix pipe), so the last, partly filled round
// costs less (launch_edge_ring picks).
template <int NWV>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void k_edge_ring(const EdgeArgs a) {
    constexpr int NT = 4, L = 128, PC = 16384;
    constexpr int W = 8;
    constexpr int BUF = Rg<W>::BUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    copy_to_lds16(wl, a.split[2], PC, true);                         // hi of W1e, W2, W3
    copy_to_lds16(wl + PC, a.split[0], PC, true);
    copy_to_lds16(wl + 2 * PC, a.split[1], PC, true);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
    u32x4* ringbase = reinterpret_cast<u32x4*>(wl + 3 * PC);                    // three window buffers (48 KiB)
    float* tb = reinterpret_cast<float*>(ringbase + 3 * BUF);
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    RingSrc src;
    {
        const u32x4* g[3] = {reinterpret_cast<const u32x4*>(a.split[2]), reinterpret_cast<const u32x4*>(a.split[0]),
                             reinterpret_cast<const u32x4*>(a.split[1])};
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            src.mid[l] = g[l] + 2048;
            src.lo[l] = g[l] + 4096;
        }
#pragma unroll
        for (int w = 0; w < 2; ++w)                                   // windows 0 and 1 of layer 1
#pragma unroll
            for (int i = 0; i < 8 / NWV; ++i) {
                ringbase[w * BUF + i * NWV * 64 + tid] = src.mid[0][w * W * 64 + i * NWV * 64 + tid];
                ringbase[w * BUF + W * 64 + i * NWV * 64 + tid] = src.lo[0][w * W * 64 + i * NWV * 64 + tid];
            }
    }
    __syncthreads();
    const u32x4* l1h = reinterpret_cast<const u32x4*>(wl);
    const u32x4* l2h = reinterpret_cast<const u32x4*>(wl + PC);
    const u32x4* l3h = reinterpret_cast<const u32x4*>(wl + 2 * PC);
    // lock-step: every wave of the block runs as many tiles as its wave 0 (the longest walk); padding tiles compute, store nothing
    TileWalk tw0(a.ntiles, 0), tw(a.ntiles, wave);
    if (tw0.tile >= tw0.end) return;
#ifndef MGN_RING_PHASES
#define MGN_RING_PHASES 2
#endif
#ifndef MGN_RING_L3_WRAP
#define MGN_RING_L3_WRAP 0       // 1: layer 3's refill without a side buffer (sp_layer_ring WRAP)
#endif
#ifndef MGN_RING_QPSTREAM
#define MGN_RING_QPSTREAM 0      // 1: P[s] / Q[r] rows stream through layer 1 (sp_layer_ring QP): no Q gather / side buffer in the turnover
#endif
#ifndef MGN_RING_ENEXT_TAIL
#define MGN_RING_ENEXT_TAIL 0    // where the last two k-steps of the next tile's e are requested: 0 = at the start of the epilogue (ahead of the e stores), 1 = at its end
#endif
#ifndef MGN_RING_ENEXT
#define MGN_RING_ENEXT 1         // 1 (default): layer 3's refill fetches the NEXT tile's e, this tile's is read again in the epilogue; 2: and stored last (spills); 0: round 3's order
#endif
#ifndef MGN_RING_PHASE_UNITS
#define MGN_RING_PHASE_UNITS 10
#endif
    const int iters = (tw0.end - tw0.tile + tw0.stride - 1) / tw0.stride;
    // Blocks of one launch start together and keep the same period, so the memory phases (epilogues) of all CUs coincide.  Long
    // launches start every second block of an XCD half a period (10 x 4 096 cycles) late: 3.454 -> 3.413 ms on M-1M (four or eight
    // groups: the same; a quarter period: nothing) -- the bursts are not what binds the epilogue (docs/experiments.md).
    if (MGN_RING_PHASES > 1 && iters >= 32) {
        const int ph = (int)(blockIdx.x / NUM_XCD) % MGN_RING_PHASES;
        for (int i = 0; i < ph * MGN_RING_PHASE_UNITS; ++i) __builtin_amdgcn_s_sleep(64);
    }
    const int last = a.tile0 + tw0.tile + (iters - 1) * tw0.stride;   // a tile that exists (loads of padding tiles go there)
    tw.tile += a.tile0;
    tw.end += a.tile0;
    auto clamp = [&](int t) { return t < tw.end ? t : last; };
    f32x16 acc[NT], y[NT];
    EdgeIdx ix = load_edge_idx_nb(a.snd, a.rcv, a.E, clamp(tw.tile), lane0 & 31);
    {
        const int h0 = lane0 >> 5;
#if !MGN_RING_QPSTREAM
        load_frag<NT>(acc, prow_ptr(a.Q, ix.r >= 0 ? ix.r : 0, L, h0), STRIDE_PROW);
#endif
        load_frag<NT>(y, tile_ptr(a.Elat, clamp(tw.tile), L, lane0), STRIDE_TILE);
    }
    f32x4 side[4];               // refill side buffer (sp_layer_ring): the first two k-steps of the P rows / of the e tile
#if !MGN_RING_QPSTREAM
    {
        const f32x4* p0 = prow_ptr(a.P, ix.s, L, lane0 >> 5);
#pragma unroll
        for (int m = 0; m < 4; ++m) side[m] = p0[m * STRIDE_PROW];
    }
#endif
    int stamp_tile = 0;
    (void)stamp_tile;
    for (int j = 0; j < iters; ++j, ++stamp_tile) {
        OPAQUE_LANE();
        const bool on = tw.tile < tw.end;
        const int tile = clamp(tw.tile);
        const int nxt = clamp(tw.tile + tw.stride);
        const EdgeIdx ixn = load_edge_idx_nb(a.snd, a.rcv, a.E, nxt, c);
        const bool valid = on && ix.r >= 0;
        const int r = ix.r >= 0 ? ix.r : 0;
        f32x4* etile = tile_ptr(a.Elat, tile, L, lane);
        u32x4* ring = ringbase + lane;
#ifdef MGN_WHATIF          // diagnostic builds (wrong results): which memory stream costs what
        const f32x4* etile_rd = (MGN_WHATIF & 1) ? tile_ptr(a.Elat, a.tile0 + wave, L, lane) : etile;
        const int ps_row = (MGN_WHATIF & 4) ? (lane & 31) : ix.s;
#else
        const f32x4* etile_rd = etile;
        const int ps_row = ix.s;
#endif
        STAMP(0);
        __builtin_amdgcn_s_setprio(0);
        RingFrag nx = ring_first<W, 0>(l1h, ring, lane);
        // layer 1 (edge part): y = e tile in, P[s] out (acc entered with Q[r], which carries b1)
#if MGN_RING_QPSTREAM
        zero_frag<NT>(acc);
        sp_layer_ring<W, 0, false, 0, 0, NWV, false, true>(acc, y, l1h, l2h, ring, src, nx, lane, tid, nullptr, nullptr, nullptr,
                                                          prow_ptr(a.P, ps_row, L, h), prow_ptr(a.Q, r, L, h));
#else
        sp_layer_ring<W, 0, false, STRIDE_PROW, 0, NWV>(acc, y, l1h, l2h, ring, src, nx, lane, tid, prow_ptr(a.P, ps_row, L, h), side);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] += y[t];
#endif
        CST(1);
        tab_frag<NT>(y, tb + T_B2 * L, h);
        CST(2);
#if MGN_RING_L3_WRAP || MGN_RING_ENEXT
        sp_layer_ring<W, 1, true, 0, 0, NWV>(y, acc, l2h, l3h, ring, src, nx, lane, tid);   // layer 2 (ReLU folded into the split)
#else
        sp_layer_ring<W, 1, true, 0, STRIDE_TILE, NWV>(y, acc, l2h, l3h, ring, src, nx, lane, tid, nullptr, side, etile_rd);   // layer 2 (ReLU folded into the split)
#endif
        CST(3);
        tab_frag<NT>(acc, tb + T_B3 * L, h);
        CST(4);
        // layer 3: y = layer 2's output in, the e tile (for the residual) out
#if MGN_RING_ENEXT
        // ... or the NEXT tile's e: nothing of the turnover then waits behind this tile's e stores (s_waitcnt vmcnt retires in order)
#if defined(MGN_WHATIF) && (MGN_WHATIF & 8)
        sp_layer_ring<W, 2, true, STRIDE_TILE, 0, NWV, true>(acc, y, l3h, l1h, ring, src, nx, lane, tid, tile_ptr(a.Elat, a.tile0 + wave, L, lane));
#else
        sp_layer_ring<W, 2, true, STRIDE_TILE, 0, NWV, true>(acc, y, l3h, l1h, ring, src, nx, lane, tid, tile_ptr(a.Elat, nxt, L, lane));
#endif
#elif MGN_RING_L3_WRAP
        sp_layer_ring<W, 2, true, STRIDE_TILE, 0, NWV, true>(acc, y, l3h, l1h, ring, src, nx, lane, tid, etile_rd);
        ring_load_e_tail(y, etile_rd);
#else
        sp_layer_ring<W, 2, true, STRIDE_TILE, 0, NWV>(acc, y, l3h, l1h, ring, src, nx, lane, tid, etile_rd, side);
#endif
        CST(5);
        EST(1);
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_PRIO);
#if MGN_RING_ENEXT
        f32x16 er[NT];               // this tile's e again, for the residual (y holds the next tile's): arrives during the LayerNorm
        ring_load_e(er, etile_rd);
#if MGN_RING_ENEXT_TAIL == 0
        ring_load_e_tail(y, tile_ptr(a.Elat, nxt, L, lane));         // k-steps 6 and 7 of the next tile's e
#endif
#endif
        {   // LayerNorm (layer_norm_frag of frag.hpp in four slices): acc = e'
            constexpr float invL = 1.0f / 128;
            float sm = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) sm += acc[t][k];
            sm += __shfl_xor(sm, 32, 64);
            const float mean = sm * invL;
            float q = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float d = acc[t][k] - mean;
                    acc[t][k] = d;
                    q += d * d;
                }
            q += __shfl_xor(q, 32, 64);
            const float rstd = ln_rstd_at(q * invL, tb + T_LN * L);
            const f32x4* g4 = reinterpret_cast<const f32x4*>(tb + T_GAMMA * L) + h;
            const f32x4* b4 = reinterpret_cast<const f32x4*>(tb + T_BETA * L) + h;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 gv = g4[2 * (4 * t + g)];
                    const f32x4 bv = b4[2 * (4 * t + g)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = acc[t][4 * g + i] * rstd * gv[i] + bv[i];
                }
            }
        }
        CST(6);
        EST(2);
#if MGN_RING_ENEXT
#pragma unroll
        for (int t = 0; t < NT; ++t) er[t] += acc[t];                // e <- e + e'
#if MGN_RING_ENEXT == 1
#if defined(MGN_WHATIF) && (MGN_WHATIF & 2)
        if (valid && a.E < 0) ring_store_e(etile, er);
#elif defined(MGN_WHATIF) && (MGN_WHATIF & 64)          // the e stores issued, but into one tile per wave (no HBM traffic)
        if (valid) ring_store_e(tile_ptr(a.Elat, a.tile0 + wave, L, lane), er);
#else
        if (valid) ring_store_e(etile, er);                          // padding rows / tiles store nothing
#endif
#endif
        CST(7);
        EST(3);
#else
#pragma unroll
        for (int t = 0; t < NT; ++t) y[t] += acc[t];                 // e <- e + e'
#if defined(MGN_WHATIF) && (MGN_WHATIF & 2)
        if (valid && a.E < 0) store_frag<NT>(etile, STRIDE_TILE, y);
#else
        if (valid) ring_store_e(etile, y);                           // padding rows / tiles store nothing
#endif
        CST(7);
        EST(3);
#if defined(MGN_WHATIF) && (MGN_WHATIF & 8)
        load_frag<NT>(y, tile_ptr(a.Elat, a.tile0 + wave, L, lane), STRIDE_TILE);
#else
        ring_load_e(y, tile_ptr(a.Elat, nxt, L, lane));              // the next tile's e, ahead of everything else of the turnover
#endif
#endif
        const int reff = ix.r >= 0 ? r : (-4 - c);
        const int rprev = __shfl_up(reff, 1, 32);
        const int rnext = __shfl_down(reff, 1, 32);
        const bool head = (c == 0) || (reff != rprev);
        const unsigned hm = (unsigned)__ballot(head);
        const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
        const int st_in = max(start, c & 16);
        const bool c1 = (c - 1 >= st_in), c2 = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
        const bool cx = (c >= 16) && (start <= 15);
        EST(4);
        PHASE_FENCE();
        asm volatile("s_nop 1");
        RG_SCAN_LEVEL(acc, c1, "row_shr:1 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c2, "row_shr:2 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c4, "row_shr:4 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c8, "row_shr:8 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, cx, "row_bcast:15 row_mask:0xa bank_mask:0xf");
        PHASE_FENCE();
        EST(5);
        const bool tail = valid && ((c == 31) || (reff != rnext));
        const int r_first = __builtin_amdgcn_readfirstlane(reff);
        const bool sl = (start == 0) && (ix.r_before == r_first);
        const bool sr = (c == 31) && (ix.r_after == reff);
        const bool to_carry = sl || sr;
        f32x4* dst = to_carry ? prow_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), L, h) : tile_ptr(a.AGG, r >> 5, L, 32 * h + (r & 31));
#if defined(MGN_WHATIF) && (MGN_WHATIF & 32)
        if (tail && a.E < 0) store_frag<NT>(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, acc);
#elif defined(MGN_WHATIF) && (MGN_WHATIF & 128)         // aggregate rows stored row-major (4 lines per row instead of 32)
        if (tail) store_frag<NT>(to_carry ? dst : prow_ptr(a.AGG, r, L, h), STRIDE_PROW, acc);
#else
        if (tail) store_frag<NT>(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, acc);
#endif
        EST(6);
        PHASE_FENCE();
        // turnover: the next tile's layer-1 accumulator starts from Q[r] (P[s] arrives during the layer)
#if MGN_RING_QPSTREAM
        // (nothing to request: the rows stream through layer 1)
#elif defined(MGN_WHATIF) && (MGN_WHATIF & 16)
        load_frag<NT>(acc, prow_ptr(a.Q, lane & 31, L, h), STRIDE_PROW);
#else
        load_frag<NT>(acc, prow_ptr(a.Q, ixn.r >= 0 ? ixn.r : 0, L, h), STRIDE_PROW);
#endif
#if MGN_RING_REFILL_AT_REQUEST != 0 && (MGN_RING_SIDE_EARLY & 2) && !MGN_RING_QPSTREAM
        {
            const f32x4* pn = prow_ptr(a.P, ixn.s, L, h);
#pragma unroll
            for (int m = 0; m < 4; ++m) side[m] = pn[m * STRIDE_PROW];
        }
#endif
#if MGN_RING_ENEXT && MGN_RING_ENEXT_TAIL == 1
        ring_load_e_tail(y, tile_ptr(a.Elat, nxt, L, lane));         // k-steps 6 and 7 of the next tile's e: first needed late in its layer 1
#endif
#if MGN_RING_ENEXT == 2
        PHASE_FENCE();
        if (valid) ring_store_e(etile, er);                          // the e stores last: every request of the turnover is older than they are
#endif
        EST(7);
        ix = ixn;
        tw.tile += tw.stride;
    }
}

// ================================================================================================
// k_edge_ring_h (round 5; MGN_FP32_SPLIT = 1, the default): k_edge_ring with every fp32 operand as TWO fp16 pieces and THREE piece
// products per (k-step, output block) -- 288 v_mfma_f32_32x32x16_f16 per tile instead of 576 bf16 ones (split_common.hpp: why that is
// still fp32 arithmetic).  Same lock-step ring, turnover, LayerNorm, scan, stores; what changes:
//   * LDS: the hi pieces of W1e, W2, W3 resident (96 KiB); the lo piece of the layer in progress through three window buffers of W
//     steps (W = 8: 8 KiB each, 12 windows and barriers per tile; every thread fetches ONE 16-byte fragment per window);
//   * every operand is scaled by a power of two so that its lo piece is a normal fp16: the weights per chunk on the host
//     (EdgeArgs.h2_s / h2_rs), the activations per ROW here -- lane = row, so the scale is one register and the row's output column
//     is un-scaled by it exactly.  Layer 1 takes the row maximum of the e tile; layer 2 that of layer 1's finished output; layer 3 a
//     bound, c2 max(acc, 0) + max(b2, 0), because layer 2's bias and un-scaling are folded into layer 3's split (FIN = 2): its
//     accumulators are never "finished" as a fragment;
//   * accumulators start from zero (the first MFMA of a block takes the constant), except layer 1's: Q[r] scaled INTO the
//     accumulator's units (64 multiplications; P[s] is added by the FMA that un-scales), since nothing has 64 registers for Q.
// ================================================================================================
#ifndef MGN_RINGH_AHEAD
#define MGN_RINGH_AHEAD 2        // windows between a window's request and its first use: 2 (default) = three buffers, stored in the window it is
#endif                           // requested in (k_edge_ring's scheme); 3 = four buffers, stored one window LATER (below): 2.401 vs 2.408 ms, nothing
template <int W>
struct Rh {
    static constexpr int WPL = 32 / W;          // windows per layer
    static constexpr int NW = 3 * WPL;          // windows per tile
    static constexpr int NB = MGN_RINGH_AHEAD + 1;   // window buffers; NW is a multiple of NB: window -> buffer is the same for every tile
    static constexpr int BUF = W * 64;          // u32x4 elements per window buffer: [step][lane] of the lo piece
    static_assert(NW % NB == 0 && NW % 2 == 0, "static window -> buffer / slot mapping");
};
// s_waitcnt vmcnt retires in order: the LDS store of a window waits for its request AND for every older load -- the refill requests of
// the window before (the next tile's e, straight from HBM), which then have 7 .. 13 steps to come back before they stall the chain
// (stamps: the layers with a refill take twice the cycles of layer 2).  AHEAD = 3: a window is requested THREE windows ahead and
// stored in the window AFTER the one it was requested in (two requests in flight per thread, four buffers): the refills get 15 .. 21 steps.
// Measured: no difference -- what the refills cost is their issue into a memory pipeline that is still draining the tile's stores.
template <int LPT> struct RhPend { u32x4 v[2][LPT]; };
struct RhFrag {
    u32x4 h, l;                                 // fragments of the next step (read one step ahead)
};
template <int W, int LYR>
DEVINL RhFrag rh_first(const u32x4* hi, const u32x4* ring, int lane) {
    constexpr int b = (Rh<W>::WPL * LYR) % Rh<W>::NB;
    RhFrag f;
    f.h = hi[lane];
    f.l = ring[b * Rh<W>::BUF];
    return f;
}
struct RhSrc {
    const u32x4* lo[3];                         // global lo pieces of layers 1..3 (W1e, W2, W3)
};
// One L x L layer (sp_layer_ring with two pieces).  sx: the row's scale; FIN / cfin / btab: see h2_split_pair (btab = the bias table of
// the layer BEFORE, lane half's offset included).  Refill: schedule 2 of sp_layer_ring (one request every second step; WRAP: no
// rotation, the last two k-steps' pieces are the caller's).
// RFPOL >= 0 (WRAP refill only): the refill's requests as buffer loads with these cache-policy bits (rfb: the tile's descriptor; lane offset
// 16 lane, piece m at m KiB) instead of plain loads through `rf`.
template <int W, int LYR, int FIN, int RFS = 0, int NWV = 8, bool WRAP = false, int RFPOL = -1>
DEVINL void h2_layer_ring(f32x16 (&acc)[4], f32x16 (&in)[4], const u32x4* hi, const u32x4* hi_next, u32x4* ring, const RhSrc& src,
                          RhFrag& nx, RhPend<W / NWV>& pend, int lane, int tid, float sx, float cfin = 0.f, const float* btab = nullptr,
                          const f32x4* rf = nullptr, const N16Buf* rfb = nullptr) {
    constexpr int ROT = 2;
    constexpr int WPL = Rh<W>::WPL, NW = Rh<W>::NW, BUF = Rh<W>::BUF, NB = Rh<W>::NB, AHEAD = MGN_RINGH_AHEAD;
    f32x4 side[2 * ROT];
    if constexpr (!WRAP && RFS > 0) {
#pragma unroll
        for (int m = 0; m < 2 * ROT; ++m) side[m] = rf[m * RFS];
    }
    // bias pair of (k-step sn, pair u): registers 8 (sn & 1) + 2 u, + 1 of block sn >> 1 (pack_tab's order)
    auto bias = [&](int sn, int u) {
        f32x2 b = {0.f, 0.f};
        if constexpr (FIN == 2) b = *reinterpret_cast<const f32x2*>(btab + 8 * (4 * (sn >> 1) + 2 * (sn & 1) + (u >> 1)) + 2 * (u & 1));
        return b;
    };
    unsigned ph[4], pl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const f32x2 b = bias(0, u);
        h2_split_pair<FIN>(ph[u], pl[u], in[0][2 * u], in[0][2 * u + 1], sx, cfin, b[0], b[1]);
    }
    constexpr int LPT = W / NWV;                 // fragments per thread in a window (NWV waves share the loading)
    unsigned voff = (unsigned)tid * 16u;
    asm volatile("" : "+v"(voff));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        unsigned nh[4], nl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int it = 4 * s + t;
            const int gw = WPL * LYR + it / W;                        // global window of this step
            const u32x4 a1 = nx.h, a2 = nx.l;
#ifndef MGN_WHATIF_FREE        // (MGN_WHATIF_FREE = n: timing-only build, wrong results -- no ring traffic, no barriers, waves 4..7 start n x 8 k cycles late)
            if (it % W == 0) {                                         // request window gw + AHEAD
                const int g2 = (gw + AHEAD) % NW, l2 = g2 / WPL, w2 = g2 % WPL;
#pragma unroll
                for (int i = 0; i < LPT; ++i)
                    pend.v[AHEAD == 2 ? 0 : gw % 2][i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src.lo[l2] + w2 * W * 64 + i * NWV * 64) + voff);
            }
#endif
            if constexpr (RFS > 0) {
                if ((t & 1) && s < 8 - ROT) {                          // registers of k-step s (free since the step began), half t >> 1
                    f32x4 v;
                    if constexpr (WRAP && RFPOL >= 0) v = n16_ld_pol<RFPOL>(*rfb, (unsigned)lane * 16u, (2 * s + (t >> 1)) * 1024);
                    else v = rf[(2 * (WRAP ? s : s + ROT) + (t >> 1)) * RFS];
#pragma unroll
                    for (int i = 0; i < 4; ++i) in[s >> 1][8 * (s & 1) + 4 * (t >> 1) + i] = v[i];
                }
            }
            if (it + 1 < 32) {
                const int gn = WPL * LYR + (it + 1) / W;
                nx.h = hi[(it + 1) * 64 + lane];
                nx.l = ring[(gn % NB) * BUF + ((it + 1) % W) * 64];
            } else if (LYR < 2) {
                nx = rh_first<W, (LYR + 1) % 3>(hi_next, ring, lane);   // (that window was written two windows ago)
            }
#ifndef MGN_WHATIF_FREE
            if (it % W == W - 2) {                                     // store window gw + 2 (AHEAD 3: requested in window gw - 1): its buffer was
                const int b2 = (gw + 2) % NB;                          // last read as window gw + 2 - NB
                const int slot = AHEAD == 2 ? 0 : (gw + 1) % 2;      // (AHEAD 2: a window is stored in the window it is requested in -- one slot)
#pragma unroll
                for (int i = 0; i < LPT; ++i) ring[b2 * BUF + i * NWV * 64 + tid - lane] = pend.v[slot][i];
            }
#endif
            if (s < 7) {
                const int sn = s + 1;
                const f32x2 b = bias(sn, t);
                h2_split_pair<FIN>(nh[t], nl[t], in[sn >> 1][8 * (sn & 1) + 2 * t], in[sn >> 1][8 * (sn & 1) + 2 * t + 1], sx, cfin, b[0], b[1]);
            }
            const sp_f16x8 bh = h2_op(ph), bl = h2_op(pl);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a2), bh, acc[t], 0, 0, 0);      // small terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a1), bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a1), bh, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef MGN_WHATIF_FREE
            if (it % W == W - 1) ring_barrier();                       // window closed: every wave has read it, window gw + 2 is in LDS
#endif
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ph[u] = nh[u];
            pl[u] = nl[u];
        }
    }
    if constexpr (RFS > 0 && !WRAP) {                                  // un-rotate: k-step u's pieces sit in the registers of k-step u - 2
        f32x16 r[4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r[u >> 1][8 * (u & 1) + j] = u < ROT ? side[2 * u + (j >> 2)][j & 3] : in[(u - ROT) >> 1][8 * ((u - ROT) & 1) + j];
#pragma unroll
        for (int t = 0; t < 4; ++t) in[t] = r[t];
    }
}

#ifndef MGN_RINGH_PHASE_UNITS
#define MGN_RINGH_PHASE_UNITS 6     // half a period of k_edge_ring_h in 4 096-cycle units (k_edge_ring: 10)
#endif
#ifndef MGN_RINGH_ESTORE_LAST
#define MGN_RINGH_ESTORE_LAST 0   // (1 spills 64 registers in the epilogue: not run) 1: this tile's e stores behind the next tile's Q request (0: right behind the residual, k_edge_ring's order)
#endif
#ifndef MGN_RING_EST_BUF
#define MGN_RING_EST_BUF 0       // k_edge_ring_h (with MGN_RINGH_STORE_INTERLEAVE): the e stores as buffer stores with these policy bits (17 = sc0 sc1: write-through);
                                 // 0: plain global stores.  As compiler-visible instructions write-through is worth nothing (2.299 -> 2.295 ms): the 1.5 % the
                                 // inline-assembly form showed came with a lost reproducibility test (MGN_RING_ESTORE above)
#endif
#ifndef MGN_RING_ENEXT_POL
#define MGN_RING_ENEXT_POL 0     // k_edge_ring_h: the next tile's e requests as buffer loads (a scalar descriptor + one lane offset: no 64-bit address arithmetic
                                 // inside layer 3) with these cache-policy bits (1 sc0, 2 nt, 16 sc1; 0 = plain: 2.337 -> 2.304 ms, the policies add nothing); -1: global loads
#endif
#ifndef MGN_RING_ER_POL
#define MGN_RING_ER_POL -1       // ... of the second read of this tile's e
#endif
#ifndef MGN_RINGH_STORE_INTERLEAVE
#define MGN_RINGH_STORE_INTERLEAVE 1   // 1: a block's residual + e stores right behind its LayerNorm (0: all sixteen stores behind the LayerNorm)
#endif
#ifndef MGN_RINGH_SCAN_SKIP
#define MGN_RINGH_SCAN_SKIP 1     // the scan's row_shr:8 level behind a wave-uniform branch (taken only by tiles with a receiver run of nine edges or more inside a 16-lane row)
#endif
#ifndef MGN_RINGH_W
#define MGN_RINGH_W 8            // steps per window of k_edge_ring_h (8: 12 barriers per tile; 16: 6)
#endif
template <int NWV>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void k_edge_ring_h(const EdgeArgs a) {
    constexpr int NT = 4, L = 128, PC = 16384;
    constexpr int W = MGN_RINGH_W;
    constexpr int BUF = Rh<W>::BUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    copy_to_lds16(wl, a.splith[2], PC, true);                        // hi of W1e, W2, W3
    copy_to_lds16(wl + PC, a.splith[0], PC, true);
    copy_to_lds16(wl + 2 * PC, a.splith[1], PC, true);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
    u32x4* ringbase = reinterpret_cast<u32x4*>(wl + 3 * PC);                    // three window buffers
    float* tb = reinterpret_cast<float*>(ringbase + Rh<W>::NB * BUF);
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    RhSrc src;
    RhPend<W / NWV> pend;
    {
        const u32x4* g[3] = {reinterpret_cast<const u32x4*>(a.splith[2]), reinterpret_cast<const u32x4*>(a.splith[0]),
                             reinterpret_cast<const u32x4*>(a.splith[1])};
#pragma unroll
        for (int l = 0; l < 3; ++l) src.lo[l] = g[l] + 2048;
#pragma unroll
        for (int w = 0; w < 2; ++w)                                   // windows 0 and 1 of layer 1
#pragma unroll
            for (int i = 0; i < W / NWV; ++i) ringbase[w * BUF + i * NWV * 64 + tid] = src.lo[0][w * W * 64 + i * NWV * 64 + tid];
#pragma unroll
        for (int i = 0; i < W / NWV; ++i) {                           // AHEAD 3: window 2 waits in the slot that window 0 stores from
            pend.v[1][i] = src.lo[(2 * W) / 32][((2 * W) % 32) * 64 + i * NWV * 64 + tid];
            pend.v[0][i] = pend.v[1][i];
        }
    }
    __syncthreads();
    const u32x4* l1h = reinterpret_cast<const u32x4*>(wl);
    const u32x4* l2h = reinterpret_cast<const u32x4*>(wl + PC);
    const u32x4* l3h = reinterpret_cast<const u32x4*>(wl + 2 * PC);
    // chunk scales (layer order): s = what the host multiplied the chunk by, rs = 1 / s
    const float sw1 = a.h2_s[2], rsw1 = a.h2_rs[2], rsw2 = a.h2_rs[0], rsw3 = a.h2_rs[1], b2pos = a.h2_b2pos;
    // lock-step: every wave of the block runs as many tiles as its wave 0 (the longest walk); padding tiles compute, store nothing
    TileWalk tw0(a.ntiles, 0), tw(a.ntiles, wave);
    if (tw0.tile >= tw0.end) return;
    const int iters = (tw0.end - tw0.tile + tw0.stride - 1) / tw0.stride;
    if (MGN_RING_PHASES > 1 && iters >= 32) {                         // (k_edge_ring: every second block of an XCD half a period late)
        const int ph = (int)(blockIdx.x / NUM_XCD) % MGN_RING_PHASES;
        for (int i = 0; i < ph * MGN_RINGH_PHASE_UNITS; ++i) __builtin_amdgcn_s_sleep(64);
    }
    const int last = a.tile0 + tw0.tile + (iters - 1) * tw0.stride;   // a tile that exists (loads of padding tiles go there)
    tw.tile += a.tile0;
    tw.end += a.tile0;
    auto clamp = [&](int t) { return t < tw.end ? t : last; };
    f32x16 acc[NT], y[NT];
    EdgeIdx ix = load_edge_idx_nb(a.snd, a.rcv, a.E, clamp(tw.tile), lane0 & 31);
    {
        const int h0 = lane0 >> 5;
        load_frag<NT>(acc, prow_ptr(a.Q, ix.r >= 0 ? ix.r : 0, L, h0), STRIDE_PROW);
        load_frag<NT>(y, tile_ptr(a.Elat, clamp(tw.tile), L, lane0), STRIDE_TILE);
    }
    int stamp_tile = 0;
    (void)stamp_tile;
#ifdef MGN_WHATIF_FREE
    if (wave >= NWV / 2)
        for (int i = 0; i < MGN_WHATIF_FREE; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    for (int j = 0; j < iters; ++j, ++stamp_tile) {
        OPAQUE_LANE();
        const bool on = tw.tile < tw.end;
        const int tile = clamp(tw.tile);
        const int nxt = clamp(tw.tile + tw.stride);
        const EdgeIdx ixn = load_edge_idx_nb(a.snd, a.rcv, a.E, nxt, c);
        const bool valid = on && ix.r >= 0;
        const int r = ix.r >= 0 ? ix.r : 0;
        f32x4* etile = tile_ptr(a.Elat, tile, L, lane);
        u32x4* ring = ringbase + lane;
        STAMP(0);
        __builtin_amdgcn_s_setprio(0);
        RhFrag nx = rh_first<W, 0>(l1h, ring, lane);
        // layer 1 (edge part): y = e tile in, P[s] out; acc enters with Q[r] (which carries b1), put into the accumulator's units
        const H2Scale x1 = h2_scale(h2_rowmax<true>(y));
        {
            const float cinv = x1.s * sw1;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[t][k] *= cinv;
        }
#if defined(MGN_WHATIF_H) && (MGN_WHATIF_H & 4)      // diagnostic builds (wrong results): which memory stream of this kernel costs what
        const int ps_row = lane & 31;
#else
        const int ps_row = ix.s;
#endif
        h2_layer_ring<W, 0, 0, STRIDE_PROW, NWV>(acc, y, l1h, l2h, ring, src, nx, pend, lane, tid, x1.s, 0.f, nullptr, prow_ptr(a.P, ps_row, L, h));
        {
            const float c1 = x1.rs * rsw1;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[t][k] = __builtin_fmaf(acc[t][k], c1, y[t][k]);
        }
        CST(1);
        const H2Scale x2 = h2_scale(h2_rowmax<false>(acc));
        zero_frag<NT>(y);
        CST(2);
        h2_layer_ring<W, 1, 1, 0, NWV>(y, acc, l2h, l3h, ring, src, nx, pend, lane, tid, x2.s);   // layer 2 (ReLU folded into the split)
        CST(3);
        const float c2 = x2.rs * rsw2;
        const H2Scale x3 = h2_scale(__builtin_fmaf(h2_rowmax<false>(y), c2, b2pos));
        zero_frag<NT>(acc);
        CST(4);
        // layer 3: y = layer 2's accumulators in (bias, un-scaling and ReLU in the split), the NEXT tile's e out
#if defined(MGN_WHATIF_H) && (MGN_WHATIF_H & 16)
        const f32x4* enext = tile_ptr(a.Elat, a.tile0 + wave, L, lane);
#else
        const f32x4* enext = tile_ptr(a.Elat, nxt, L, lane);
#endif
#if MGN_RING_ENEXT_POL >= 0
        const N16Buf enb = n16_buf(a.Elat + (int64_t)nxt * (TILE * L), TILE * L * 4);      // (nxt is wave-uniform)
        h2_layer_ring<W, 2, 2, STRIDE_TILE, NWV, true, MGN_RING_ENEXT_POL>(acc, y, l3h, l1h, ring, src, nx, pend, lane, tid, x3.s, c2, tb + T_B2 * L + 4 * h, enext, &enb);
#else
        h2_layer_ring<W, 2, 2, STRIDE_TILE, NWV, true>(acc, y, l3h, l1h, ring, src, nx, pend, lane, tid, x3.s, c2, tb + T_B2 * L + 4 * h, enext);
#endif
        CST(5);
        EST(1);
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_PRIO);
        f32x16 er[NT];               // this tile's e again, for the residual (y holds the next tile's): arrives during the LayerNorm
#if MGN_RING_EST_BUF
        const N16Buf esb = n16_buf(a.Elat + (int64_t)tile * (TILE * L), TILE * L * 4);
#endif
#if defined(MGN_WHATIF_H) && (MGN_WHATIF_H & 1)
#pragma unroll
        for (int t = 0; t < NT; ++t) er[t] = acc[t];
#else
#if MGN_RING_ER_POL >= 0
        {
            const N16Buf erb = n16_buf(a.Elat + (int64_t)tile * (TILE * L), TILE * L * 4);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const f32x4 v = n16_ld_pol<MGN_RING_ER_POL>(erb, (unsigned)lane * 16u, m * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) er[m >> 2][4 * (m & 3) + i] = v[i];
            }
        }
#else
        ring_load_e(er, etile);
#endif
#endif
#if MGN_RING_ENEXT_POL >= 0
#pragma unroll
        for (int m = 12; m < 16; ++m) {
            const f32x4 v = n16_ld_pol<MGN_RING_ENEXT_POL>(enb, (unsigned)lane * 16u, m * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) y[3][4 * (m & 3) + i] = v[i];
        }
#else
        ring_load_e_tail(y, enext);                                  // k-steps 6 and 7 of the next tile's e
#endif
        {   // bias + un-scaling of layer 3, then LayerNorm: acc = e'
            constexpr float invL = 1.0f / 128;
            const float c3 = x3.rs * rsw3;
            const f32x4* b34 = reinterpret_cast<const f32x4*>(tb + T_B3 * L) + h;
            float sm = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = b34[2 * (4 * t + g)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = __builtin_fmaf(acc[t][4 * g + i], c3, bv[i]);
                        acc[t][4 * g + i] = v;
                        sm += v;
                    }
                }
            sm += __shfl_xor(sm, 32, 64);
            const float mean = sm * invL;
            float q = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float d = acc[t][k] - mean;
                    acc[t][k] = d;
                    q += d * d;
                }
            q += __shfl_xor(q, 32, 64);
            const float rstd = ln_rstd_at(q * invL, tb + T_LN * L);
            const f32x4* g4 = reinterpret_cast<const f32x4*>(tb + T_GAMMA * L) + h;
            const f32x4* b4 = reinterpret_cast<const f32x4*>(tb + T_BETA * L) + h;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 gv = g4[2 * (4 * t + g)];
                    const f32x4 bv = b4[2 * (4 * t + g)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = acc[t][4 * g + i] * rstd * gv[i] + bv[i];
                }
#if MGN_RINGH_STORE_INTERLEAVE && !MGN_RINGH_ESTORE_LAST && !defined(MGN_WHATIF_H)
                // the block's residual and its four stores right behind its LayerNorm: the stores of the first blocks are on their way while the
                // others are still normalised
                er[t] += acc[t];
#if MGN_RING_EST_BUF
                {   // as buffer stores (compiler-visible instructions, unlike ring_store_e_piece's inline assembly: hazards and counters are the compiler's)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = er[t][4 * g + i];
                        if (valid) n16_st_pol<MGN_RING_EST_BUF>(esb, (unsigned)lane * 16u, (4 * t + g) * 1024, v);
                    }
                }
#else
                if (valid) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = er[t][4 * g + i];
                        ring_store_e_piece(etile + (4 * t + g) * STRIDE_TILE, v);
                    }
                }
#endif
#endif
            }
        }
        CST(6);
        EST(2);
#if !(MGN_RINGH_STORE_INTERLEAVE && !MGN_RINGH_ESTORE_LAST && !defined(MGN_WHATIF_H))
#pragma unroll
        for (int t = 0; t < NT; ++t) er[t] += acc[t];                // e <- e + e'
#endif
#if !MGN_RINGH_ESTORE_LAST && !(MGN_RINGH_STORE_INTERLEAVE && !defined(MGN_WHATIF_H))
#if defined(MGN_WHATIF_H) && (MGN_WHATIF_H & 2)
        if (valid && a.E < 0) ring_store_e(etile, er);
#else
        if (valid) ring_store_e(etile, er);                          // padding rows / tiles store nothing
#endif
#endif
        CST(7);
        EST(3);
        const int reff = ix.r >= 0 ? r : (-4 - c);
        const int rprev = __shfl_up(reff, 1, 32);
        const int rnext = __shfl_down(reff, 1, 32);
        const bool head = (c == 0) || (reff != rprev);
        const unsigned hm = (unsigned)__ballot(head);
        const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
        const int st_in = max(start, c & 16);
        const bool c1 = (c - 1 >= st_in), c2s = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
        const bool cx = (c >= 16) && (start <= 15);
        EST(4);
        PHASE_FENCE();
        asm volatile("s_nop 1");
        RG_SCAN_LEVEL(acc, c1, "row_shr:1 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c2s, "row_shr:2 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c4, "row_shr:4 row_mask:0xf bank_mask:0xf");
#if MGN_RINGH_SCAN_SKIP
        // receiver runs of nine edges and more are rare on a mesh (M-1M: none): their level only where a lane of the tile needs it
        if (__builtin_amdgcn_ballot_w64(c8) != 0)
#endif
        {
            RG_SCAN_LEVEL(acc, c8, "row_shr:8 row_mask:0xf bank_mask:0xf");
        }
        RG_SCAN_LEVEL(acc, cx, "row_bcast:15 row_mask:0xa bank_mask:0xf");
        PHASE_FENCE();
        EST(5);
        const bool tail = valid && ((c == 31) || (reff != rnext));
        const int r_first = __builtin_amdgcn_readfirstlane(reff);
        const bool sl = (start == 0) && (ix.r_before == r_first);
        const bool sr = (c == 31) && (ix.r_after == reff);
        const bool to_carry = sl || sr;
        f32x4* dst = to_carry ? prow_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), L, h) : tile_ptr(a.AGG, r >> 5, L, 32 * h + (r & 31));
#if defined(MGN_WHATIF_H) && (MGN_WHATIF_H & 32)
        if (tail && a.E < 0) store_frag<NT>(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, acc);
#else
        if (tail) store_frag_pol<NT, MGN_AGG_STORE>(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, acc);
#endif
        EST(6);
        PHASE_FENCE();
        // turnover: the next tile's layer-1 accumulator starts from Q[r] (P[s] arrives during the layer)
#if defined(MGN_WHATIF_H) && (MGN_WHATIF_H & 64)
        zero_frag<NT>(acc);                                          // (no Q request at all: what the wait for it behind the stores costs)
#elif defined(MGN_WHATIF_H) && (MGN_WHATIF_H & 8)
        load_frag<NT>(acc, prow_ptr(a.Q, lane & 31, L, h), STRIDE_PROW);
#else
        load_frag<NT>(acc, prow_ptr(a.Q, ixn.r >= 0 ? ixn.r : 0, L, h), STRIDE_PROW);
#endif
#if MGN_RINGH_ESTORE_LAST
        // the e stores LAST: s_waitcnt vmcnt retires in order and counts stores, so everything the next tile waits for first (its Q rows)
        // is requested ahead of them; they drain under the next tile's layer 1
        PHASE_FENCE();
#if defined(MGN_WHATIF_H) && (MGN_WHATIF_H & 2)
        if (valid && a.E < 0) ring_store_e(etile, er);
#else
        if (valid) ring_store_e(etile, er);                          // padding rows / tiles store nothing
#endif
#endif
        EST(7);
        ix = ixn;
        tw.tile += tw.stride;
    }
}

// ================================================================================================
// k_edge_ring_hs (round 6): k_edge_ring_h with NO resident weight piece -- hi and lo of the layer in progress both pass through the
// window ring (windows of HS_W = 4 steps: 8 KiB = one 16-byte fragment per thread, three buffers, 24 windows and barriers per tile;
// 24 KiB of L2 weight traffic per tile and wave instead of 12) -- and the 128 KiB of LDS this frees hold the tile's e: every wave
// parks its e tile (16 KiB, its own region: no synchronisation) when the tile starts and takes it back for the residual.  e is read
// from memory ONCE per step (k_edge_ring_h: twice, 3.07 GB per launch on M-1M through the Infinity Cache), the epilogue's burst of
// requests loses its sixteen largest, and a block's LDS prologue is 28 KiB instead of 150.
// ================================================================================================
constexpr int HS_W = 4;
#ifndef MGN_HS_PREMUL
#define MGN_HS_PREMUL 0          // 1: the gathered Q rows are put into the accumulator's units block by block inside layer 1's first k-step (measured: no gain, 2.174 vs 2.165 ms)
#endif
#ifndef MGN_HS_REQ_STEP
#define MGN_HS_REQ_STEP 0        // step of window gw (0 .. 3) at which window gw + MGN_HS_AHEAD is requested
#endif
#ifndef MGN_HS_AHEAD
#define MGN_HS_AHEAD 4           // 3: two requests in flight (a request 7 steps before its LDS store), 4: three (11 steps).  M-1M: 2.336 (k_edge_ring_h) ->
                                 // 2.340 with the request in the window of its store -> 2.288 / 2.260 / 2.245 / 2.250 at AHEAD 3 (request at step 2 / 0), 4, 5
#endif
template <int NCH>                              // NCH: chains (L x L products) per tile, all through the one ring
struct RsT {
    static constexpr int W = HS_W;
    static constexpr int WPL = 32 / W;          // windows per chain: 8
    static constexpr int NW = NCH * WPL;        // windows per tile
    static constexpr int NB = 3;                // window buffers (a window is requested two windows ahead and stored in the window it is requested in)
    static constexpr int BUF = 2 * W * 64;      // u32x4 elements per buffer: [step][hi, lo][lane]
    static constexpr int AHEAD = MGN_HS_AHEAD;  // windows between a window's request and its first use
    static constexpr int SLOTS = AHEAD - 1;     // requests in flight per thread
    static_assert(NW % NB == 0 && NW % SLOTS == 0 && AHEAD >= 3, "static window -> buffer / slot mapping");
};
typedef RsT<3> Rs;                              // k_edge_ring_hs: three layers
template <int LYR>                              // (window -> buffer is window % NB for every NCH: the first window of chain LYR)
DEVINL RhFrag rs_first(const u32x4* ring) {
    constexpr int b = (Rs::WPL * LYR) % Rs::NB;
    RhFrag f;
    f.h = ring[b * Rs::BUF];
    f.l = ring[b * Rs::BUF + 64];
    return f;
}
template <int NCH>
struct RsSrcT {
    const u32x4* w[NCH];                        // the chunks of the tile's chains in their order: hi piece at + 0, lo piece at + 2048 fragments
};
typedef RsSrcT<3> RsSrc;
// element e of window w of a chunk (e = (step * 2 + piece) * 64 + lane): where it sits in the chunk's global pieces
DEVINL const u32x4* rs_src(const u32x4* chunk, int w, int e) { return chunk + ((e >> 6) & 1) * 2048 + (w * Rs::W + (e >> 7)) * 64 + (e & 63); }
// One L x L layer with both pieces from the ring.  Otherwise h2_layer_ring (same split, same products, same order; refill as there).
// LYR: the chain's place among the NCH chains of a tile (the ring's schedule); WRAP refill: through the descriptor rfb.
// PREMUL: acc enters in other units (the gathered Q rows): block t is multiplied by `premul` at step (0, t), just ahead of its first product --
// its four pieces are waited for there, not all sixteen before the chain.
template <int LYR, int FIN, int RFS = 0, int NWV = 8, bool WRAP = false, int NCH = 3, bool PREMUL = false>
DEVINL void hs_layer_ring(f32x16 (&acc)[4], f32x16 (&in)[4], u32x4* ring, const RsSrcT<NCH>& src, RhFrag& nx, u32x4 (&pend)[Rs::SLOTS][Rs::BUF / (NWV * 64)], int lane,
                          int tid, float sx, float cfin = 0.f, const float* btab = nullptr, const f32x4* rf = nullptr, const N16Buf* rfb = nullptr, int rfs_rt = 0, float premul = 1.f) {
    constexpr int ROT = 2;
    constexpr int W = Rs::W, WPL = Rs::WPL, NW = RsT<NCH>::NW, BUF = Rs::BUF, NB = Rs::NB;
    constexpr int LPT = BUF / (NWV * 64);        // fragments per thread in a window
    f32x4 side[2 * ROT];
    if constexpr (!WRAP && RFS > 0) {
#pragma unroll
        for (int m = 0; m < 2 * ROT; ++m) side[m] = rf[m * RFS];
    }
    auto bias = [&](int sn, int u) {
        f32x2 b = {0.f, 0.f};
        if constexpr (FIN == 2) b = *reinterpret_cast<const f32x2*>(btab + 8 * (4 * (sn >> 1) + 2 * (sn & 1) + (u >> 1)) + 2 * (u & 1));
        return b;
    };
    unsigned ph[4], pl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const f32x2 b = bias(0, u);
        h2_split_pair<FIN>(ph[u], pl[u], in[0][2 * u], in[0][2 * u + 1], sx, cfin, b[0], b[1]);
    }
    // the thread's place inside a window, as a byte offset from the window's first hi fragment (one register, kept opaque: as 64-bit
    // addresses per window hipcc hoists 24 of them out of the tile loop and spills)
    unsigned voff[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int e = i * NWV * 64 + tid;
        voff[i] = (unsigned)((((e >> 6) & 1) * 2048 + (e >> 7) * 64 + (e & 63)) * 16);
        asm volatile("" : "+v"(voff[i]));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        unsigned nh[4], nl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int it = 4 * s + t;
            const int gw = WPL * LYR + it / W;                        // global window of this step
            const u32x4 a1 = nx.h, a2 = nx.l;
            // window gw + AHEAD is requested at step MGN_HS_REQ_STEP of window gw and goes to LDS at the last step of window gw + AHEAD - 2 (into
            // the buffer that window gw + AHEAD - 3 has just closed on): AHEAD - 1 requests in flight, one register slot each
            if (it % W == MGN_HS_REQ_STEP) {
                const int g2 = (gw + Rs::AHEAD) % NW, l2 = g2 / WPL, w2 = g2 % WPL;
#pragma unroll
                for (int i = 0; i < LPT; ++i)
                    pend[gw % Rs::SLOTS][i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src.w[l2] + w2 * W * 64) + voff[i]);
            }
            if constexpr (RFS != 0) {
                if ((t & 1) && s < 8 - ROT) {                          // registers of k-step s (free since the step began), half t >> 1
                    f32x4 v;
                    if constexpr (WRAP && RFS < 0) v = rf[(int64_t)(2 * s + (t >> 1)) * rfs_rt];   // (per-lane pointer and stride: the aggregate rows)
                    else if constexpr (WRAP) v = n16_ld(*rfb, (unsigned)lane * 16u, (2 * s + (t >> 1)) * 1024);
                    else v = rf[(2 * (s + ROT) + (t >> 1)) * RFS];
#pragma unroll
                    for (int i = 0; i < 4; ++i) in[s >> 1][8 * (s & 1) + 4 * (t >> 1) + i] = v[i];
                }
            }
            if (it + 1 < 32) {
                const int gn = WPL * LYR + (it + 1) / W;
                nx.h = ring[(gn % NB) * BUF + (((it + 1) % W) * 2) * 64];
                nx.l = ring[(gn % NB) * BUF + (((it + 1) % W) * 2 + 1) * 64];
            } else if (LYR < NCH - 1) {
                nx = rs_first<LYR + 1>(ring);                          // (that window was written two windows ago)
            }
            if (it % W == W - 1) {                                     // store window gw + 2 (requested in window gw + 2 - AHEAD): its buffer was last read as window gw - 1
                const int b2 = (gw + 2) % NB;
#pragma unroll
                for (int i = 0; i < LPT; ++i) ring[b2 * BUF + i * NWV * 64 + tid - lane] = pend[(gw + 2 - Rs::AHEAD + NW) % Rs::SLOTS][i];
            }
            if (s < 7) {
                const int sn = s + 1;
                const f32x2 b = bias(sn, t);
                h2_split_pair<FIN>(nh[t], nl[t], in[sn >> 1][8 * (sn & 1) + 2 * t], in[sn >> 1][8 * (sn & 1) + 2 * t + 1], sx, cfin, b[0], b[1]);
            }
            if constexpr (PREMUL) {
                if (s == 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[t][k] *= premul;
                }
            }
            const sp_f16x8 bh = h2_op(ph), bl = h2_op(pl);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a2), bh, acc[t], 0, 0, 0);      // small terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a1), bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a1), bh, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (it % W == W - 1) ring_barrier();                       // window closed: every wave has read it, window gw + 2 is in LDS
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ph[u] = nh[u];
            pl[u] = nl[u];
        }
    }
    if constexpr (RFS > 0 && !WRAP) {                                  // un-rotate: k-step u's pieces sit in the registers of k-step u - 2
        f32x16 r[4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r[u >> 1][8 * (u & 1) + j] = u < ROT ? side[2 * u + (j >> 2)][j & 3] : in[(u - ROT) >> 1][8 * ((u - ROT) & 1) + j];
#pragma unroll
        for (int t = 0; t < 4; ++t) in[t] = r[t];
    }
}
template <int NWV>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void k_edge_ring_hs(const EdgeArgs a) {
    constexpr int NT = 4, L = 128;
    constexpr int BUF = Rs::BUF, LPT = BUF / (NWV * 64);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
    u32x4* ringbase = reinterpret_cast<u32x4*>(smem);                            // three window buffers (24 KiB)
    f32x4* park = reinterpret_cast<f32x4*>(ringbase + Rs::NB * BUF) + wave * 1024;   // this wave's e tile: [piece 16][lane 64] x 16 B
    float* tb = reinterpret_cast<float*>(ringbase + Rs::NB * BUF) + NWV * 4096;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    RsSrc src;
    u32x4 pend[Rs::SLOTS][LPT];
    {
        src.w[0] = reinterpret_cast<const u32x4*>(a.splith[2]);
        src.w[1] = reinterpret_cast<const u32x4*>(a.splith[0]);
        src.w[2] = reinterpret_cast<const u32x4*>(a.splith[1]);
#pragma unroll
        for (int w = 0; w < 2; ++w)                                   // windows 0 and 1 of layer 1
#pragma unroll
            for (int i = 0; i < LPT; ++i) ringbase[w * BUF + i * NWV * 64 + tid] = *rs_src(src.w[0], w, i * NWV * 64 + tid);
#pragma unroll
        for (int sl = 0; sl < Rs::SLOTS; ++sl)
#pragma unroll
            for (int i = 0; i < LPT; ++i) pend[sl][i] = *rs_src(src.w[0], 2, i * NWV * 64 + tid);
#pragma unroll
        for (int x = 2; x < Rs::AHEAD; ++x)                           // windows 2 .. AHEAD - 1 wait in the slots their stores will read
#pragma unroll
            for (int i = 0; i < LPT; ++i) pend[(x - Rs::AHEAD + Rs::NW) % Rs::SLOTS][i] = *rs_src(src.w[x / Rs::WPL], x % Rs::WPL, i * NWV * 64 + tid);
    }
    __syncthreads();
    // chunk scales (layer order): s = what the host multiplied the chunk by, rs = 1 / s
    const float sw1 = a.h2_s[2], rsw1 = a.h2_rs[2], rsw2 = a.h2_rs[0], rsw3 = a.h2_rs[1], b2pos = a.h2_b2pos;
    // lock-step: every wave of the block runs as many tiles as its wave 0 (the longest walk); padding tiles compute, store nothing
    TileWalk tw0(a.ntiles, 0), tw(a.ntiles, wave);
    if (tw0.tile >= tw0.end) return;
    const int iters = (tw0.end - tw0.tile + tw0.stride - 1) / tw0.stride;
    if (MGN_RING_PHASES > 1 && iters >= 32) {                         // (k_edge_ring: every second block of an XCD half a period late)
        const int ph = (int)(blockIdx.x / NUM_XCD) % MGN_RING_PHASES;
        for (int i = 0; i < ph * MGN_RINGH_PHASE_UNITS; ++i) __builtin_amdgcn_s_sleep(64);
    }
    const int last = a.tile0 + tw0.tile + (iters - 1) * tw0.stride;   // a tile that exists (loads of padding tiles go there)
    tw.tile += a.tile0;
    tw.end += a.tile0;
    auto clamp = [&](int t) { return t < tw.end ? t : last; };
    f32x16 acc[NT], y[NT];
    EdgeIdx ix = load_edge_idx_nb(a.snd, a.rcv, a.E, clamp(tw.tile), lane0 & 31);
    {
        const int h0 = lane0 >> 5;
        load_frag<NT>(acc, prow_ptr(a.Q, ix.r >= 0 ? ix.r : 0, L, h0), STRIDE_PROW);
        load_frag<NT>(y, tile_ptr(a.Elat, clamp(tw.tile), L, lane0), STRIDE_TILE);
    }
    int stamp_tile = 0;
    (void)stamp_tile;
    for (int j = 0; j < iters; ++j, ++stamp_tile) {
        OPAQUE_LANE();
        const bool on = tw.tile < tw.end;
        const int tile = clamp(tw.tile);
        const int nxt = clamp(tw.tile + tw.stride);
        const EdgeIdx ixn = load_edge_idx_nb(a.snd, a.rcv, a.E, nxt, c);
        const bool valid = on && ix.r >= 0;
        const int r = ix.r >= 0 ? ix.r : 0;
        f32x4* etile = tile_ptr(a.Elat, tile, L, lane);
        u32x4* ring = ringbase + lane;
        STAMP(0);
        __builtin_amdgcn_s_setprio(0);
        RhFrag nx = rs_first<0>(ring);
#pragma unroll
        for (int m = 0; m < 16; ++m) {                               // park this tile's e for the residual (the wave's own 16 KiB of LDS)
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = y[m >> 2][4 * (m & 3) + i];
            park[m * 64 + lane] = v;
        }
        // layer 1 (edge part): y = e tile in, P[s] out; acc enters with Q[r] (which carries b1), put into the accumulator's units
        const H2Scale x1 = h2_scale(h2_rowmax<true>(y));
        const int ps_row = ix.s;
#if MGN_HS_PREMUL
        hs_layer_ring<0, 0, STRIDE_PROW, NWV, false, 3, true>(acc, y, ring, src, nx, pend, lane, tid, x1.s, 0.f, nullptr, prow_ptr(a.P, ps_row, L, h), nullptr, 0, x1.s * sw1);
#else
        {
            const float cinv = x1.s * sw1;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[t][k] *= cinv;
        }
        hs_layer_ring<0, 0, STRIDE_PROW, NWV>(acc, y, ring, src, nx, pend, lane, tid, x1.s, 0.f, nullptr, prow_ptr(a.P, ps_row, L, h));
#endif
        {
            const float c1 = x1.rs * rsw1;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[t][k] = __builtin_fmaf(acc[t][k], c1, y[t][k]);
        }
        CST(1);
        const H2Scale x2 = h2_scale(h2_rowmax<false>(acc));
        zero_frag<NT>(y);
        CST(2);
        hs_layer_ring<1, 1, 0, NWV>(y, acc, ring, src, nx, pend, lane, tid, x2.s);   // layer 2 (ReLU folded into the split)
        CST(3);
        const float c2 = x2.rs * rsw2;
        const H2Scale x3 = h2_scale(__builtin_fmaf(h2_rowmax<false>(y), c2, b2pos));
        zero_frag<NT>(acc);
        CST(4);
        // layer 3: y = layer 2's accumulators in (bias, un-scaling and ReLU in the split), the NEXT tile's e out
        const N16Buf enb = n16_buf(a.Elat + (int64_t)nxt * (TILE * L), TILE * L * 4);      // (nxt is wave-uniform)
        hs_layer_ring<2, 2, STRIDE_TILE, NWV, true>(acc, y, ring, src, nx, pend, lane, tid, x3.s, c2, tb + T_B2 * L + 4 * h, nullptr, &enb);
        CST(5);
        EST(1);
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_PRIO);
#pragma unroll
        for (int m = 12; m < 16; ++m) {                              // k-steps 6 and 7 of the next tile's e
            const f32x4 v = n16_ld(enb, (unsigned)lane * 16u, m * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) y[3][4 * (m & 3) + i] = v[i];
        }
        {   // bias + un-scaling of layer 3, then LayerNorm: acc = e'
            constexpr float invL = 1.0f / 128;
            const float c3 = x3.rs * rsw3;
            const f32x4* b34 = reinterpret_cast<const f32x4*>(tb + T_B3 * L) + h;
            float sm = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = b34[2 * (4 * t + g)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = __builtin_fmaf(acc[t][4 * g + i], c3, bv[i]);
                        acc[t][4 * g + i] = v;
                        sm += v;
                    }
                }
            sm += __shfl_xor(sm, 32, 64);
            const float mean = sm * invL;
            float q = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float d = acc[t][k] - mean;
                    acc[t][k] = d;
                    q += d * d;
                }
            q += __shfl_xor(q, 32, 64);
            const float rstd = ln_rstd_at(q * invL, tb + T_LN * L);
            const f32x4* g4 = reinterpret_cast<const f32x4*>(tb + T_GAMMA * L) + h;
            const f32x4* b4 = reinterpret_cast<const f32x4*>(tb + T_BETA * L) + h;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 gv = g4[2 * (4 * t + g)];
                    const f32x4 bv = b4[2 * (4 * t + g)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = acc[t][4 * g + i] * rstd * gv[i] + bv[i];
                }
                // the block's residual -- this tile's e comes back from the wave's LDS region -- and its four stores right behind its LayerNorm
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = park[(4 * t + g) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += acc[t][4 * g + i];
                    if (valid) etile[(4 * t + g) * STRIDE_TILE] = v;
                }
            }
        }
        CST(6);
        EST(2);
        CST(7);
        EST(3);
        const int reff = ix.r >= 0 ? r : (-4 - c);
        const int rprev = __shfl_up(reff, 1, 32);
        const int rnext = __shfl_down(reff, 1, 32);
        const bool head = (c == 0) || (reff != rprev);
        const unsigned hm = (unsigned)__ballot(head);
        const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
        const int st_in = max(start, c & 16);
        const bool c1 = (c - 1 >= st_in), c2s = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
        const bool cx = (c >= 16) && (start <= 15);
        EST(4);
        PHASE_FENCE();
        asm volatile("s_nop 1");
        RG_SCAN_LEVEL(acc, c1, "row_shr:1 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c2s, "row_shr:2 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c4, "row_shr:4 row_mask:0xf bank_mask:0xf");
        if (__builtin_amdgcn_ballot_w64(c8) != 0)                    // (receiver runs of nine edges and more: rare on a mesh)
        {
            RG_SCAN_LEVEL(acc, c8, "row_shr:8 row_mask:0xf bank_mask:0xf");
        }
        RG_SCAN_LEVEL(acc, cx, "row_bcast:15 row_mask:0xa bank_mask:0xf");
        PHASE_FENCE();
        EST(5);
        const bool tail = valid && ((c == 31) || (reff != rnext));
        const int r_first = __builtin_amdgcn_readfirstlane(reff);
        const bool sl = (start == 0) && (ix.r_before == r_first);
        const bool sr = (c == 31) && (ix.r_after == reff);
        const bool to_carry = sl || sr;
        f32x4* dst = to_carry ? prow_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), L, h) : tile_ptr(a.AGG, r >> 5, L, 32 * h + (r & 31));
        if (tail) store_frag<NT>(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, acc);
        EST(6);
        PHASE_FENCE();
        // turnover: the next tile's layer-1 accumulator starts from Q[r] (P[s] arrives during the layer)
        load_frag<NT>(acc, prow_ptr(a.Q, ixn.r >= 0 ? ixn.r : 0, L, h), STRIDE_PROW);
        EST(7);
        ix = ixn;
        tw.tile += tw.stride;
    }
}


// ================================================================================================
// Processor node step (K6) on the split path: k_node_step<4, *, false>'s tile loop with its four L x L chunks on the bf16 matrix
// cores.  split[]: the chunks in NodeArgs.chunk order (0: W2, 1: W3, 2: W1[0:L] (node part), 3: W1[L:2L] (aggregate part)), each as
// hi / mid / lo pieces.  LDS: the four hi pieces (128 KiB) + tables; mid and lo stream from L2 (sp_layer_otf rings).  The V tile is
// read again for the residual (its registers carry the aggregate and then the second layer's output meanwhile).
// TWO: a second edge set -- its aggregate is one more layer-1 block (split[6] = W1[2L:3L]); LDS is full with four hi pieces, so all
// three pieces of that chunk stream from L2 (rings four groups deep: the same 48 ring registers as two rings of six).
// ================================================================================================
// priorities of the node-side split kernels: inside the MFMA chains the older wave of a SIMD (waves 0-3) and the younger one (4-7) may
// differ, so that the pipe's arbiter pulls the two waves of a SIMD apart instead of letting them share every chain and then both wait
// in their memory phases at once
#ifndef MGN_NODE_CPRIO_OLD
#define MGN_NODE_CPRIO_OLD 0
#endif
#ifndef MGN_NODE_CPRIO_YOUNG
#define MGN_NODE_CPRIO_YOUNG 0
#endif
#ifndef MGN_NODE_MPRIO
#define MGN_NODE_MPRIO MGN_PRIO
#endif
#define NODE_CHAIN_PRIO()                                                        \
    do {                                                                         \
        if (MGN_NODE_CPRIO_OLD == MGN_NODE_CPRIO_YOUNG) __builtin_amdgcn_s_setprio(MGN_NODE_CPRIO_OLD); \
        else if (wave < 4) __builtin_amdgcn_s_setprio(MGN_NODE_CPRIO_OLD);       \
        else __builtin_amdgcn_s_setprio(MGN_NODE_CPRIO_YOUNG);                   \
    } while (0)
#ifndef MGN_SP2_D3
#define MGN_SP2_D3 4
#endif
template <bool TWO>
__global__ __launch_bounds__(512, 2) void k_node_split(const NodeArgs a) {
    constexpr int NT = 4, L = 128, PC = 16384, D = MGN_SP2_D;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    {
        const bool fast = a.ntiles <= MGN_FAST_PRELOAD_TILES;
        copy_to_lds16(wl, a.split[2], PC, fast);
        copy_to_lds16(wl + PC, a.split[3], PC, fast);
        copy_to_lds16(wl + 2 * PC, a.split[0], PC, fast);
        copy_to_lds16(wl + 3 * PC, a.split[1], PC, fast);
    }
    float* tb = smem + 4 * PC / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const u32x4* lvh = reinterpret_cast<const u32x4*>(wl);
    const u32x4* lah = reinterpret_cast<const u32x4*>(wl + PC);
    const u32x4* l2h = reinterpret_cast<const u32x4*>(wl + 2 * PC);
    const u32x4* l3h = reinterpret_cast<const u32x4*>(wl + 3 * PC);
    const u32x4* gv = reinterpret_cast<const u32x4*>(a.split[2]);      // 2048 fragments per piece
    const u32x4* ga = reinterpret_cast<const u32x4*>(a.split[3]);
    const u32x4* g2 = reinterpret_cast<const u32x4*>(a.split[0]);
    const u32x4* g3 = reinterpret_cast<const u32x4*>(a.split[1]);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stagger_second_half(wave, a.stagger);
    TileWalk tw(a.ntiles, wave, MGN_SPREAD_ROUNDS_NODE);
    if (tw.tile >= tw.end) return;
    f32x16 x[NT], acc[NT];
    load_frag<NT>(x, tile_ptr(a.V, tw.tile, L, lane0), STRIDE_TILE);
    for (;;) {
        OPAQUE_LANE();
        const int tile = tw.tile;
        const int next = tile + tw.stride;
        const bool has_next = next < tw.end;
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        f32x4* vtile = tile_ptr(a.V, tile, L, lane);
        NODE_CHAIN_PRIO();
        tab_frag<NT>(acc, tb + T_B1 * L, h);
#if defined(MGN_WHATIF_NODE) && (MGN_WHATIF_NODE & 1)   // diagnostic (wrong results): every piece from LDS
        sp_layer_otf<false, false, false, D>(acc, x, lvh, lvh, lvh, lane);
        LOAD_AGGREGATE(NT, x, a.rowptr, a.AGG, a.CARRY, a.zero_row);
        sp_layer_otf<false, false, false, D>(acc, x, lah, lah, lah, lane);
        tab_frag<NT>(x, tb + T_B2 * L, h);
        sp_layer_otf<false, false, true, D>(x, acc, l2h, l2h, l2h, lane);
        tab_frag<NT>(acc, tb + T_B3 * L, h);
        sp_layer_otf<false, false, true, D>(acc, x, l3h, l3h, l3h, lane);
#else
#if MGN_SP_CARRY
        // the weight rings are carried from chain to chain: every chain requests the first fragments of the next one in its last steps
        constexpr int O1 = 32 % D, O2 = 64 % D, O3 = 96 % D;
        SpRing<D> rg;
        sp_layer_otf<true, true, false, D, false, 0, false, true>(acc, x, lvh, gv + 2048, gv + 4096, lane, &rg, ga + 2048, ga + 4096);   // layer 1, node part
        LOAD_AGGREGATE(NT, x, a.rowptr, a.AGG, a.CARRY, a.zero_row);
        if constexpr (TWO) {
            sp_layer_otf<true, true, false, D, false, O1, true, false>(acc, x, lah, ga + 2048, ga + 4096, lane, &rg);                  // aggregate part
            const u32x4* gb = reinterpret_cast<const u32x4*>(a.split[6]);
            LOAD_AGGREGATE(NT, x, a.rowptr2, a.AGG2, a.CARRY2, a.zero_row2);
            sp_layer_otf<true, true, false, MGN_SP2_D3, true>(acc, x, gb, gb + 2048, gb + 4096, lane);                                 // the second set's aggregate
            tab_frag<NT>(x, tb + T_B2 * L, h);
            sp_layer_otf<true, true, true, D, false, 0, false, true>(x, acc, l2h, g2 + 2048, g2 + 4096, lane, &rg, g3 + 2048, g3 + 4096);   // layer 2
            tab_frag<NT>(acc, tb + T_B3 * L, h);
            sp_layer_otf<true, true, true, D, false, O1, true, false>(acc, x, l3h, g3 + 2048, g3 + 4096, lane, &rg);                   // layer 3
        } else {
            sp_layer_otf<true, true, false, D, false, O1, true, true>(acc, x, lah, ga + 2048, ga + 4096, lane, &rg, g2 + 2048, g2 + 4096);   // aggregate part
            tab_frag<NT>(x, tb + T_B2 * L, h);
            sp_layer_otf<true, true, true, D, false, O2, true, true>(x, acc, l2h, g2 + 2048, g2 + 4096, lane, &rg, g3 + 2048, g3 + 4096);    // layer 2
            tab_frag<NT>(acc, tb + T_B3 * L, h);
            sp_layer_otf<true, true, true, D, false, O3, true, false>(acc, x, l3h, g3 + 2048, g3 + 4096, lane, &rg);                  // layer 3
        }
#else
        sp_layer_otf<true, true, false, D>(acc, x, lvh, gv + 2048, gv + 4096, lane);      // layer 1, node part
        LOAD_AGGREGATE(NT, x, a.rowptr, a.AGG, a.CARRY, a.zero_row);
        sp_layer_otf<true, true, false, D>(acc, x, lah, ga + 2048, ga + 4096, lane);      // layer 1, aggregate part
        if constexpr (TWO) {                                                              // layer 1, the second edge set's aggregate
            const u32x4* gb = reinterpret_cast<const u32x4*>(a.split[6]);
            LOAD_AGGREGATE(NT, x, a.rowptr2, a.AGG2, a.CARRY2, a.zero_row2);
            sp_layer_otf<true, true, false, MGN_SP2_D3, true>(acc, x, gb, gb + 2048, gb + 4096, lane);
        }
        tab_frag<NT>(x, tb + T_B2 * L, h);
        sp_layer_otf<true, true, true, D>(x, acc, l2h, g2 + 2048, g2 + 4096, lane);       // layer 2 (ReLU folded into the split)
        tab_frag<NT>(acc, tb + T_B3 * L, h);
        sp_layer_otf<true, true, true, D>(acc, x, l3h, g3 + 2048, g3 + 4096, lane);       // layer 3
#endif
#endif
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
#if defined(MGN_WHATIF_NODE) && (MGN_WHATIF_NODE & 2)   // diagnostic: the residual reads one cached tile per wave
        load_frag<NT>(x, tile_ptr(a.V, wave, L, lane), STRIDE_TILE);
#else
        load_frag<NT>(x, vtile, STRIDE_TILE);                        // v again, for the residual
#endif
        layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);
#pragma unroll
        for (int t = 0; t < NT; ++t) x[t] += acc[t];                 // v <- v + v'
#if MGN_NODE_VNEXT_FIRST
        // the next tile's V is requested AHEAD of this tile's V stores, into the registers of v' (dead from here): s_waitcnt vmcnt
        // retires in order and counts stores, so a load requested behind the sixteen stores is not back before they are acknowledged
        // (k_edge_ring's turnover, docs/experiments.md round 4)
        PHASE_FENCE();
        if (has_next) load_frag<NT>(acc, tile_ptr(a.V, next, L, lane), STRIDE_TILE);
        PHASE_FENCE();
#endif
#if defined(MGN_WHATIF_NODE) && (MGN_WHATIF_NODE & 4)   // diagnostic: no store
        if (valid && a.n < 0) store_frag<NT>(vtile, STRIDE_TILE, x);
#else
        if (valid) store_frag<NT>(vtile, STRIDE_TILE, x);
#endif
        if (!has_next) break;
        PHASE_FENCE();
#if MGN_NODE_VNEXT_FIRST
#pragma unroll
        for (int t = 0; t < NT; ++t) x[t] = acc[t];
#else
        load_frag<NT>(x, tile_ptr(a.V, next, L, lane), STRIDE_TILE);
#endif
        tw.tile = next;
    }
}

// P, Q projection of the next step (k_project) on the split path.  split[4] = WP, split[5] = WQ; LDS: hi + mid of both (128 KiB),
// the lo pieces stream.  Tiles [tile0, tile0 + ntiles).
__global__ __launch_bounds__(512, 2) void k_project_split(const NodeArgs a) {
    constexpr int NT = 4, L = 128, PC = 16384, D1 = MGN_SP2_D1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    {
        const bool fast = a.ntiles <= MGN_FAST_PRELOAD_TILES;
        copy_to_lds16(wl, a.split[4], 2 * PC, fast);                 // hi + mid are adjacent
        copy_to_lds16(wl + 2 * PC, a.split[5], 2 * PC, fast);
    }
    float* tb = smem + 4 * PC / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const u32x4* lph = reinterpret_cast<const u32x4*>(wl);
    const u32x4* lpm = reinterpret_cast<const u32x4*>(wl + PC);
    const u32x4* lqh = reinterpret_cast<const u32x4*>(wl + 2 * PC);
    const u32x4* lqm = reinterpret_cast<const u32x4*>(wl + 3 * PC);
    const u32x4* gp = reinterpret_cast<const u32x4*>(a.split[4]);
    const u32x4* gq = reinterpret_cast<const u32x4*>(a.split[5]);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TileWalk tw(a.ntiles, wave, MGN_SPREAD_ROUNDS_NODE);
    if (tw.tile >= tw.end) return;
    f32x16 x[NT], acc[NT];
    load_frag<NT>(x, tile_ptr(a.V, a.tile0 + tw.tile, L, lane0), STRIDE_TILE);
    for (;;) {
        OPAQUE_LANE();
        const int tile = a.tile0 + tw.tile;
        const int next = tw.tile + tw.stride;
        const bool has_next = next < tw.end;
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        NODE_CHAIN_PRIO();
        zero_frag<NT>(acc);
#if defined(MGN_WHATIF_NODE) && (MGN_WHATIF_NODE & 1)
        sp_layer_otf<false, false, false, D1>(acc, x, lph, lpm, lpm, lane);
#else
#if MGN_SP_CARRY
        SpRing<D1> rg;                                               // WP's chain requests the first lo fragments of WQ in its last steps
        sp_layer_otf<false, true, false, D1, false, 0, false, true>(acc, x, lph, lpm, gp + 4096, lane, &rg, nullptr, gq + 4096);
#else
        sp_layer_otf<false, true, false, D1>(acc, x, lph, lpm, gp + 4096, lane);
#endif
#endif
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
#if defined(MGN_WHATIF_NODE) && (MGN_WHATIF_NODE & 8)      // diagnostic: P / Q stored tile-major (coalesced) instead of row-major
        if (valid) store_frag<NT>(tile_ptr(a.P, tile, L, lane), STRIDE_TILE, acc);
#elif defined(MGN_WHATIF_NODE) && (MGN_WHATIF_NODE & 16)   // diagnostic: no P / Q stores
        if (valid && a.n < 0) store_frag<NT>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, acc);
#else
        if (valid) store_frag<NT>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, acc);
#endif
        NODE_CHAIN_PRIO();
        tab_frag<NT>(acc, tb + T_BQ * L, h);
#if defined(MGN_WHATIF_NODE) && (MGN_WHATIF_NODE & 1)
        sp_layer_otf<false, false, false, D1>(acc, x, lqh, lqm, lqm, lane);
#else
#if MGN_SP_CARRY
        sp_layer_otf<false, true, false, D1, false, 32 % D1, true, false>(acc, x, lqh, lqm, gq + 4096, lane, &rg);
#else
        sp_layer_otf<false, true, false, D1>(acc, x, lqh, lqm, gq + 4096, lane);
#endif
#endif
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
#if MGN_NODE_VNEXT_FIRST
        PHASE_FENCE();
        if (has_next) load_frag<NT>(x, tile_ptr(a.V, a.tile0 + next, L, lane), STRIDE_TILE);   // ahead of the Q stores (x is dead: both projections read it)
        PHASE_FENCE();
#endif
#if defined(MGN_WHATIF_NODE) && (MGN_WHATIF_NODE & 8)
        if (valid) store_frag<NT>(tile_ptr(a.Q, tile, L, lane), STRIDE_TILE, acc);
#elif defined(MGN_WHATIF_NODE) && (MGN_WHATIF_NODE & 16)
        if (valid && a.n < 0) store_frag<NT>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, acc);
#else
        if (valid) store_frag<NT>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, acc);
#endif
        if (!has_next) break;
#if !MGN_NODE_VNEXT_FIRST
        PHASE_FENCE();
        load_frag<NT>(x, tile_ptr(a.V, a.tile0 + next, L, lane), STRIDE_TILE);
#endif
        tw.tile = next;
    }
}

// ================================================================================================
// Node side on two fp16 pieces (round 5): k_node_split_h, k_project_split_h -- k_node_split / k_project_split with three piece
// products per step (split_common.hpp; the scaling rules of k_edge_ring_h).  One layer = h2_layer_otf: hi piece LDS-resident, lo piece
// LDS-resident too (GL = false: the projection, whose four pieces fill 128 KiB) or streamed from L2 through ONE per-wave register
// ring D fragments deep, carried from chain to chain (k_node_split_h: four hi pieces resident; 128 KiB of lo fragments per tile and
// wave where the bf16 pieces stream 256).
// Units: an accumulator of a chain holds (true value) x (row scale) x (chunk scale).  The node MLP's first layer has two input blocks
// with their own row scales (v and the aggregate): the first chain's result is finished (bias, true units) and then put into the
// second chain's units, 2 x 64 VALU instructions; layer 2's split takes the RAW accumulators of layer 1 -- a ReLU and a power of two
// commute -- with the row maximum taken on the raw values, so nothing is finished there.
// ================================================================================================
#ifndef MGN_SPH_D
#define MGN_SPH_D 8              // depth of the lo-piece ring of k_node_split_h
#endif
#ifndef MGN_NODE_NEXT_FIRST
#define MGN_NODE_NEXT_FIRST 0    // k_node_split_h: the next tile's v requested ahead of this tile's stores (the residual's sum goes to the other array)
#endif
#ifndef MGN_PROJ_VREFILL
#define MGN_PROJ_VREFILL 0       // k_project_split_h: k-steps of the Q chain's input refilled with the next tile's v inside the chain
#endif
#ifndef MGN_NRH_AGG_REFILL
#define MGN_NRH_AGG_REFILL 0     // k_node_ring_hs: 1 = the aggregate rows requested inside the first chain (built, parity green, no gain: 0.810 vs 0.806 ms --
                                 // the phase costs its bytes, 0.6 GB, not its latency); 0: between the chains, LOAD_AGGREGATE as it is
#endif
#ifndef MGN_NODE_VBUF
#define MGN_NODE_VBUF 0          // k_node_split_h: its v loads (the refill inside layer 3, the next tile's v) as buffer loads through the tile's descriptor
#endif
#ifndef MGN_NODE_VREFILL
#define MGN_NODE_VREFILL 6       // k_node_split_h: k-steps of layer 3's input refilled with v (for the residual) inside the chain; 0: v requested after the chain
#endif
__global__ __launch_bounds__(512, 2) void k_node_split_h(const NodeArgs a) {
    constexpr int NT = 4, L = 128, PC = 16384, D = MGN_SPH_D;
    static_assert(32 % D == 0, "the carried ring keeps offset 0 from chain to chain");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    {
        const bool fast = a.ntiles <= MGN_FAST_PRELOAD_TILES;
        copy_to_lds16(wl, a.splith[2], PC, fast);
        copy_to_lds16(wl + PC, a.splith[3], PC, fast);
        copy_to_lds16(wl + 2 * PC, a.splith[0], PC, fast);
        copy_to_lds16(wl + 3 * PC, a.splith[1], PC, fast);
    }
    float* tb = smem + 4 * PC / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const u32x4* lvh = reinterpret_cast<const u32x4*>(wl);
    const u32x4* lah = reinterpret_cast<const u32x4*>(wl + PC);
    const u32x4* l2h = reinterpret_cast<const u32x4*>(wl + 2 * PC);
    const u32x4* l3h = reinterpret_cast<const u32x4*>(wl + 3 * PC);
    const u32x4* gv = reinterpret_cast<const u32x4*>(a.splith[2]) + 2048;      // the lo pieces (2048 fragments behind the hi piece)
    const u32x4* ga = reinterpret_cast<const u32x4*>(a.splith[3]) + 2048;
    const u32x4* g2 = reinterpret_cast<const u32x4*>(a.splith[0]) + 2048;
    const u32x4* g3 = reinterpret_cast<const u32x4*>(a.splith[1]) + 2048;
    const float rswv = a.h2_rs[2], swa = a.h2_s[3], rswa = a.h2_rs[3], rsw2 = a.h2_rs[0], rsw3 = a.h2_rs[1], b2pos = a.h2_b2pos;
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stagger_second_half(wave, a.stagger);
    TileWalk tw(a.ntiles, wave, MGN_SPREAD_ROUNDS_NODE);
    if (tw.tile >= tw.end) return;
    f32x16 x[NT], acc[NT];
    load_frag<NT>(x, tile_ptr(a.V, tw.tile, L, lane0), STRIDE_TILE);
    int stamp_tile = 0;
    (void)stamp_tile;
    for (;; ++stamp_tile) {
        OPAQUE_LANE();
        STAMP(0);
        const int tile = tw.tile;
        const int next = tile + tw.stride;
        const bool has_next = next < tw.end;
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        f32x4* vtile = tile_ptr(a.V, tile, L, lane);
        NODE_CHAIN_PRIO();
        SpRingH<D> rg;
        const H2Scale sv = h2_scale(h2_rowmax<true>(x));
        zero_frag<NT>(acc);
#ifdef MGN_WHATIF_NODEH          // timing-only build (wrong results): every lo piece read from LDS (the hi piece's bytes): what the L2 stream costs
        h2_layer_otf<false, 0, 1>(acc, x, lvh, lvh, lane, sv.s);
#else
        h2_layer_otf<true, 0, D, 0, false, true>(acc, x, lvh, gv, lane, sv.s, 0.f, nullptr, &rg, ga);            // layer 1, node part
#endif
        STAMP(1);
        LOAD_AGGREGATE(NT, x, a.rowptr, a.AGG, a.CARRY, a.zero_row);
        h2_finish_frag<NT>(acc, sv.rs * rswv, tb + T_B1 * L, h);                                                  // true units, b1 in
        const H2Scale sa = h2_scale(h2_rowmax<true>(x));
        h2_scale_frag<NT>(acc, sa.s * swa);                                                                       // the aggregate chain's units
        STAMP(2);
#ifdef MGN_WHATIF_NODEH
        h2_layer_otf<false, 0, 1>(acc, x, lah, lah, lane, sa.s);
#else
        h2_layer_otf<true, 0, D, 0, true, true>(acc, x, lah, ga, lane, sa.s, 0.f, nullptr, &rg, g2);             // layer 1, aggregate part
#endif
        STAMP(3);
        const H2Scale s2 = h2_scale(h2_rowmax<false>(acc));                                                       // (on the raw accumulators)
        zero_frag<NT>(x);
#ifdef MGN_WHATIF_NODEH
        h2_layer_otf<false, 1, 1>(x, acc, l2h, l2h, lane, s2.s);
#else
        h2_layer_otf<true, 1, D, 0, true, true>(x, acc, l2h, g2, lane, s2.s, 0.f, nullptr, &rg, g3);             // layer 2
#endif
        STAMP(4);
        const float c2 = s2.rs * rsw2 * (sa.rs * rswa);
        const H2Scale s3 = h2_scale(__builtin_fmaf(h2_rowmax<false>(x), c2, b2pos));
        zero_frag<NT>(acc);
#ifdef MGN_WHATIF_NODEH
        h2_layer_otf<false, 2, 1, 0, false, false, 6, STRIDE_TILE>(acc, x, l3h, l3h, lane, s3.s, c2, tb + T_B2 * L + 4 * h, nullptr, nullptr, vtile);
        STAMP(5);
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
        h2_load_tail<6, STRIDE_TILE>(x, vtile);
#elif MGN_NODE_VREFILL
        // layer 3; the registers of its input are refilled, as the split releases them, with v again (for the residual)
#if MGN_NODE_VBUF
        const N16Buf vb = n16_buf(a.V + (int64_t)tile * (TILE * L), TILE * L * 4);          // (tile is wave-uniform)
        h2_layer_otf<true, 2, D, 0, true, false, MGN_NODE_VREFILL, STRIDE_TILE, true>(acc, x, l3h, g3, lane, s3.s, c2, tb + T_B2 * L + 4 * h, &rg, nullptr, vtile, &vb);
        STAMP(5);
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
        h2_load_tile_buf<2 * MGN_NODE_VREFILL>(x, vb, lane);
#else
        h2_layer_otf<true, 2, D, 0, true, false, MGN_NODE_VREFILL, STRIDE_TILE>(acc, x, l3h, g3, lane, s3.s, c2, tb + T_B2 * L + 4 * h, &rg, nullptr, vtile);
        STAMP(5);
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
        h2_load_tail<MGN_NODE_VREFILL, STRIDE_TILE>(x, vtile);
#endif
#else
        h2_layer_otf<true, 2, D, 0, true, false>(acc, x, l3h, g3, lane, s3.s, c2, tb + T_B2 * L + 4 * h, &rg);   // layer 3
        STAMP(5);
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
        load_frag<NT>(x, vtile, STRIDE_TILE);                        // v again, for the residual
#endif
        h2_finish_frag<NT>(acc, s3.rs * rsw3, tb + T_B3 * L, h);
        layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);
#if MGN_NODE_NEXT_FIRST
        // the next tile's v is requested AHEAD of this tile's stores (s_waitcnt vmcnt retires in order and counts stores): it lands in acc's
        // registers and moves over at the end of the tile (64 v_mov: hipcc spilled ~80 registers when the sum went to acc instead)
#pragma unroll
        for (int t = 0; t < NT; ++t) x[t] += acc[t];                 // v <- v + v'
        STAMP(6);
        PHASE_FENCE();
        load_frag<NT>(acc, tile_ptr(a.V, has_next ? next : tile, L, lane), STRIDE_TILE);   // (the last tile requests itself: no branch around the request)
        PHASE_FENCE();
        if (valid) store_frag_pol<NT, MGN_NODE_STORE>(vtile, STRIDE_TILE, x);
        STAMP(7);
        if (!has_next) break;
        PHASE_FENCE();
#pragma unroll
        for (int t = 0; t < NT; ++t) x[t] = acc[t];
#else
#pragma unroll
        for (int t = 0; t < NT; ++t) x[t] += acc[t];                 // v <- v + v'
        STAMP(6);
        if (valid) store_frag_pol<NT, MGN_NODE_STORE>(vtile, STRIDE_TILE, x);
        STAMP(7);
        if (!has_next) break;
        PHASE_FENCE();
#if MGN_NODE_VBUF
        {
            const N16Buf nb = n16_buf(a.V + (int64_t)next * (TILE * L), TILE * L * 4);
            h2_load_tile_buf<0>(x, nb, lane);
        }
#else
        load_frag<NT>(x, tile_ptr(a.V, next, L, lane), STRIDE_TILE);
#endif
#endif
        tw.tile = next;
    }
}

// ================================================================================================
// k_node_ring_hs (round 6): node MLP + residual + P / Q projection of the NEXT step in ONE lock-step launch, every weight piece (six
// chunks, hi and lo) through the window ring of k_edge_ring_hs, the freed LDS holding the tile's v for the residual.  Against
// k_node_split_h + k_project_split_h: v is read once instead of three times (the residual's second read and the projection's read are
// gone: 3.79 -> 2.6 GB per step on M-1M), one launch and one LDS prologue (28 KiB) instead of two (2 x 132 KiB).  The arithmetic of
// the two kernels in their order: the same bits.  Chains of a tile: W1 node part, W1 aggregate part, W2, W3, WP, WQ (48 windows).
// ================================================================================================
__global__ __launch_bounds__(512, 2) void k_node_ring_hs(const NodeArgs a) {
    constexpr int NT = 4, L = 128, NWV = 8, NCH = 6;
    typedef RsT<NCH> R6;
    constexpr int BUF = Rs::BUF, LPT = BUF / (NWV * 64);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
    u32x4* ringbase = reinterpret_cast<u32x4*>(smem);
    f32x4* park = reinterpret_cast<f32x4*>(ringbase + Rs::NB * BUF) + wave * 1024;   // this wave's v tile
    float* tb = reinterpret_cast<float*>(ringbase + Rs::NB * BUF) + NWV * 4096;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    RsSrcT<NCH> src;
    u32x4 pend[Rs::SLOTS][LPT];
    {
        const int order[NCH] = {2, 3, 0, 1, 4, 5};                    // NodeArgs.chunk numbering of the chains
#pragma unroll
        for (int i = 0; i < NCH; ++i) src.w[i] = reinterpret_cast<const u32x4*>(a.splith[order[i]]);
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int i = 0; i < LPT; ++i) ringbase[w * BUF + i * NWV * 64 + tid] = *rs_src(src.w[0], w, i * NWV * 64 + tid);
#pragma unroll
        for (int sl = 0; sl < Rs::SLOTS; ++sl)
#pragma unroll
            for (int i = 0; i < LPT; ++i) pend[sl][i] = *rs_src(src.w[0], 2, i * NWV * 64 + tid);
#pragma unroll
        for (int x = 2; x < Rs::AHEAD; ++x)
#pragma unroll
            for (int i = 0; i < LPT; ++i) pend[(x - Rs::AHEAD + R6::NW) % Rs::SLOTS][i] = *rs_src(src.w[x / Rs::WPL], x % Rs::WPL, i * NWV * 64 + tid);
    }
    __syncthreads();
    const float rswv = a.h2_rs[2], swa = a.h2_s[3], rswa = a.h2_rs[3], rsw2 = a.h2_rs[0], rsw3 = a.h2_rs[1], b2pos = a.h2_b2pos;
    const float rswp = a.h2_rs[4], rswq = a.h2_rs[5];
    // lock-step: every wave of the block runs as many tiles as its wave 0; padding tiles compute on the last tile's rows and store nothing
    TileWalk tw0(a.ntiles, 0, MGN_SPREAD_ROUNDS_NODE), tw(a.ntiles, wave, MGN_SPREAD_ROUNDS_NODE);
    if (tw0.tile >= tw0.end) return;
    const int iters = (tw0.end - tw0.tile + tw0.stride - 1) / tw0.stride;
    const int last = tw0.tile + (iters - 1) * tw0.stride;
    auto clamp = [&](int t) { return t < tw.end ? t : last; };
    f32x16 x[NT], acc[NT];
    load_frag<NT>(x, tile_ptr(a.V, clamp(tw.tile), L, lane0), STRIDE_TILE);
    // the receiver CSR of the tile's rows, requested a tile ahead (the aggregate rows they point to are requested inside the first chain)
    int ra0, ra1;
    {
        const int n0 = clamp(tw.tile) * TILE + (lane0 & 31);
        ra0 = n0 < a.n ? a.rowptr[n0] : 0;
        ra1 = n0 < a.n ? a.rowptr[n0 + 1] : 0;
    }
    for (int j = 0; j < iters; ++j) {
        OPAQUE_LANE();
        const bool on = tw.tile < tw.end;
        const int tile = clamp(tw.tile);
        const int nxt = clamp(tw.tile + tw.stride);
        const int n = tile * TILE + c;
        const bool valid = on && n < a.n;
        const int nn = (n < a.n) ? n : 0;
        u32x4* ring = ringbase + lane;
        __builtin_amdgcn_s_setprio(0);
        RhFrag nx = rs_first<0>(ring);
#pragma unroll
        for (int m = 0; m < 16; ++m) {                               // park v for the residual
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = x[m >> 2][4 * (m & 3) + i];
            park[m * 64 + lane] = v;
        }
        const H2Scale sv = h2_scale(h2_rowmax<true>(x));
        zero_frag<NT>(acc);
#if MGN_NRH_AGG_REFILL
        // LOAD_AGGREGATE (tile_common.hpp) in two parts: where a row's sum lives is known from the CSR requested a tile ahead; its sixteen pieces
        // are requested inside the chain, into the registers of v as the split releases them; straddling runs add their other carry rows behind it
        const int T1 = ra0 >> 5, T2 = (ra1 - 1) >> 5;
        const int extra = (ra1 > ra0 && T2 > T1) ? (T2 - T1) : 0;
        const bool from_agg = (ra1 > ra0) && !extra;
        const f32x4* src0 = from_agg ? tile_ptr(a.AGG, tile, L, lane) : prow_ptr(a.CARRY, extra ? (int64_t)(2 * T1 + 1) : a.zero_row, L, h);
        const int astride = from_agg ? STRIDE_TILE : STRIDE_PROW;
        hs_layer_ring<0, 0, -1, NWV, true, NCH>(acc, x, ring, src, nx, pend, lane, tid, sv.s, 0.f, nullptr, src0, nullptr, astride);   // layer 1, node part
#pragma unroll
        for (int m = 12; m < 16; ++m) {
            const f32x4 v = src0[(int64_t)m * astride];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[3][4 * (m & 3) + i] = v[i];
        }
        if (__any(extra >= 1)) add_frag<NT>(x, prow_ptr(a.CARRY, extra >= 1 ? (int64_t)2 * (T1 + 1) : a.zero_row, L, h), STRIDE_PROW);
        if (__any(extra >= 2))
            for (int q = 2; __any(q <= extra); ++q)
                if (q <= extra) add_frag<NT>(x, prow_ptr(a.CARRY, (int64_t)2 * (T1 + q), L, h), STRIDE_PROW);
#else
        hs_layer_ring<0, 0, 0, NWV, false, NCH>(acc, x, ring, src, nx, pend, lane, tid, sv.s);                    // layer 1, node part
        {
            const bool valid_row = n < a.n;                          // (LOAD_AGGREGATE reads `valid`, `nn`, `tile`: rows, not whether the tile stores)
            const bool valid = valid_row;
#if defined(MGN_WHATIF_NRH) && (MGN_WHATIF_NRH & 1)      // timing-only builds (wrong results): what each memory phase of this kernel costs
            (void)valid;
#else
            LOAD_AGGREGATE(NT, x, a.rowptr, a.AGG, a.CARRY, a.zero_row);
#endif
        }
#endif
        h2_finish_frag<NT>(acc, sv.rs * rswv, tb + T_B1 * L, h);                                                  // true units, b1 in
        const H2Scale sa = h2_scale(h2_rowmax<true>(x));
        h2_scale_frag<NT>(acc, sa.s * swa);                                                                       // the aggregate chain's units
        hs_layer_ring<1, 0, 0, NWV, false, NCH>(acc, x, ring, src, nx, pend, lane, tid, sa.s);                    // layer 1, aggregate part
        const H2Scale s2 = h2_scale(h2_rowmax<false>(acc));
        zero_frag<NT>(x);
        hs_layer_ring<2, 1, 0, NWV, false, NCH>(x, acc, ring, src, nx, pend, lane, tid, s2.s);                    // layer 2
        const float c2 = s2.rs * rsw2 * (sa.rs * rswa);
        const H2Scale s3 = h2_scale(__builtin_fmaf(h2_rowmax<false>(x), c2, b2pos));
        zero_frag<NT>(acc);
        hs_layer_ring<3, 2, 0, NWV, false, NCH>(acc, x, ring, src, nx, pend, lane, tid, s3.s, c2, tb + T_B2 * L + 4 * h);   // layer 3
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
        h2_finish_frag<NT>(acc, s3.rs * rsw3, tb + T_B3 * L, h);
        layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);
#pragma unroll
        for (int m = 0; m < 16; ++m) {                               // v <- v + v' (v back from the wave's LDS region), stored and kept for the projection
            const f32x4 v = park[m * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[m >> 2][4 * (m & 3) + i] = v[i] + acc[m >> 2][4 * (m & 3) + i];
        }
#if defined(MGN_WHATIF_NRH) && (MGN_WHATIF_NRH & 2)
        if (valid && a.n < 0) store_frag<NT>(tile_ptr(a.V, tile, L, lane), STRIDE_TILE, x);
#else
        if (valid) store_frag<NT>(tile_ptr(a.V, tile, L, lane), STRIDE_TILE, x);
#endif
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(0);
        const H2Scale sp = h2_scale(h2_rowmax<true>(x));
        zero_frag<NT>(acc);
        hs_layer_ring<4, 0, 0, NWV, false, NCH>(acc, x, ring, src, nx, pend, lane, tid, sp.s);                    // P = v W1s
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
        h2_scale_frag<NT>(acc, sp.rs * rswp);
#if defined(MGN_WHATIF_NRH) && (MGN_WHATIF_NRH & 4)
        if (valid && a.n < 0) store_frag<NT>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, acc);
#else
        if (valid) store_frag<NT>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, acc);
#endif
        __builtin_amdgcn_s_setprio(0);
        zero_frag<NT>(acc);
        // Q = v W1r + b1; its input registers are refilled, as the split releases them, with the NEXT tile's v
        const N16Buf vnb = n16_buf(a.V + (int64_t)nxt * (TILE * L), TILE * L * 4);
        hs_layer_ring<5, 0, STRIDE_TILE, NWV, true, NCH>(acc, x, ring, src, nx, pend, lane, tid, sp.s, 0.f, nullptr, nullptr, &vnb);
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
#pragma unroll
        for (int m = 12; m < 16; ++m) {                              // k-steps 6 and 7 of the next tile's v
            const f32x4 v = n16_ld(vnb, (unsigned)lane * 16u, m * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[3][4 * (m & 3) + i] = v[i];
        }
        {
            const int n1 = nxt * TILE + c;                           // the next tile's CSR entries
            ra0 = n1 < a.n ? a.rowptr[n1] : 0;
            ra1 = n1 < a.n ? a.rowptr[n1 + 1] : 0;
        }
        h2_finish_frag<NT>(acc, sp.rs * rswq, tb + T_BQ * L, h);
#if defined(MGN_WHATIF_NRH) && (MGN_WHATIF_NRH & 4)
        if (valid && a.n < 0) store_frag<NT>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, acc);
#else
        if (valid) store_frag<NT>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, acc);
#endif
        tw.tile += tw.stride;
    }
}

__global__ __launch_bounds__(512, 2) void k_project_split_h(const NodeArgs a) {
    constexpr int NT = 4, L = 128, PC = 16384;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    {
        const bool fast = a.ntiles <= MGN_FAST_PRELOAD_TILES;
        copy_to_lds16(wl, a.splith[4], 2 * PC, fast);                // hi + lo are adjacent
        copy_to_lds16(wl + 2 * PC, a.splith[5], 2 * PC, fast);
    }
    float* tb = smem + 4 * PC / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const u32x4* lph = reinterpret_cast<const u32x4*>(wl);
    const u32x4* lpl = reinterpret_cast<const u32x4*>(wl + PC);
    const u32x4* lqh = reinterpret_cast<const u32x4*>(wl + 2 * PC);
    const u32x4* lql = reinterpret_cast<const u32x4*>(wl + 3 * PC);
    const float rswp = a.h2_rs[4], rswq = a.h2_rs[5];
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TileWalk tw(a.ntiles, wave, MGN_SPREAD_ROUNDS_NODE);
    if (tw.tile >= tw.end) return;
    f32x16 x[NT], acc[NT];
    load_frag<NT>(x, tile_ptr(a.V, a.tile0 + tw.tile, L, lane0), STRIDE_TILE);
    for (;;) {
        OPAQUE_LANE();
        const int tile = a.tile0 + tw.tile;
        const int next = tw.tile + tw.stride;
        const bool has_next = next < tw.end;
        const int n = tile * TILE + c;
        const bool valid = n < a.n;
        const int nn = valid ? n : 0;
        NODE_CHAIN_PRIO();
        const H2Scale sv = h2_scale(h2_rowmax<true>(x));
        zero_frag<NT>(acc);
        h2_layer_otf<false, 0, 1>(acc, x, lph, lpl, lane, sv.s);
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
        h2_scale_frag<NT>(acc, sv.rs * rswp);
        if (valid) store_frag_pol<NT, MGN_PROJ_STORE>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, acc);
        NODE_CHAIN_PRIO();
        zero_frag<NT>(acc);
#if MGN_PROJ_VREFILL
        // the second chain's input registers are refilled, as the split releases them, with the NEXT tile's v
        const f32x4* vnext = tile_ptr(a.V, a.tile0 + (has_next ? next : tw.tile), L, lane);
        h2_layer_otf<false, 0, 1, 0, false, false, MGN_PROJ_VREFILL, STRIDE_TILE>(acc, x, lqh, lql, lane, sv.s, 0.f, nullptr, nullptr, nullptr, vnext);
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
        h2_load_tail<MGN_PROJ_VREFILL, STRIDE_TILE>(x, vnext);
        h2_finish_frag<NT>(acc, sv.rs * rswq, tb + T_BQ * L, h);
        if (valid) store_frag_pol<NT, MGN_PROJ_STORE>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, acc);
        if (!has_next) break;
#else
        h2_layer_otf<false, 0, 1>(acc, x, lqh, lql, lane, sv.s);
        __builtin_amdgcn_s_setprio(MGN_NODE_MPRIO);
        h2_finish_frag<NT>(acc, sv.rs * rswq, tb + T_BQ * L, h);
        if (valid) store_frag_pol<NT, MGN_PROJ_STORE>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, acc);
        if (!has_next) break;
        PHASE_FENCE();
        load_frag<NT>(x, tile_ptr(a.V, a.tile0 + next, L, lane), STRIDE_TILE);
#endif
        tw.tile = next;
    }
}

template <typename K, typename A>
static hipError_t sp_launch(K kern, const A& a, const LaunchCfg& lc, hipStream_t s, bool& attr_set) {
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(lc.blocks), dim3(lc.threads), lc.lds, s, a);
    return hipGetLastError();
}
hipError_t launch_edge_ring(const EdgeArgs& a, const LaunchCfg& lc, hipStream_t s) {
    static bool attr_set8 = false, attr_set4 = false;
    if (lc.threads == 256) return sp_launch(k_edge_ring<4>, a, lc, s, attr_set4);
    return sp_launch(k_edge_ring<8>, a, lc, s, attr_set8);
}
static int g_ringh_stream = [] { const char* e = getenv("MGN_RINGH_STREAM"); return e ? atoi(e) : 1; }();   // 1 (default): k_edge_ring_hs (no resident weight piece, e parked in LDS: read once); 0: k_edge_ring_h
int edge_ring_h_streamed() { return g_ringh_stream; }
hipError_t launch_edge_ring_h(const EdgeArgs& a, const LaunchCfg& lc, hipStream_t s) {
    static bool attr_set8 = false, attr_set4 = false, attr_s8 = false, attr_s4 = false;
    if (g_ringh_stream) {
        LaunchCfg ls = lc;
        ls.lds = (size_t)Rs::NB * Rs::BUF * 16 + (size_t)(lc.threads / 64) * 16384 + (size_t)T_COUNT * 128 * 4 + 64;
        if (lc.threads == 256) return sp_launch(k_edge_ring_hs<4>, a, ls, s, attr_s4);
        return sp_launch(k_edge_ring_hs<8>, a, ls, s, attr_s8);
    }
    if (lc.threads == 256) return sp_launch(k_edge_ring_h<4>, a, lc, s, attr_set4);
    return sp_launch(k_edge_ring_h<8>, a, lc, s, attr_set8);
}
size_t edge_ring_h_lds() { return (size_t)3 * 32768 + (size_t)Rh<MGN_RINGH_W>::NB * Rh<MGN_RINGH_W>::BUF * 16 + (size_t)T_COUNT * 128 * 4 + 64; }
hipError_t launch_node_split(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s) {
    static bool attr_set = false;
    static bool attr_set2 = false;
    if (a.AGG2) return sp_launch(k_node_split<true>, a, lc, s, attr_set2);
    return sp_launch(k_node_split<false>, a, lc, s, attr_set);
}
hipError_t launch_node_split_h(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s) {
    static bool attr_set = false;
    return sp_launch(k_node_split_h, a, lc, s, attr_set);
}
static int g_node_ring_hs = [] { const char* e = getenv("MGN_NODE_RING_HS"); return e ? atoi(e) : 1; }();   // 1 (default): node MLP + projection as k_node_ring_hs where both would run; 0: k_node_split_h + k_project_split_h
int node_ring_hs_enabled() { return g_node_ring_hs; }
hipError_t launch_node_ring_hs(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s) {
    static bool attr_set = false;
    LaunchCfg ls = lc;
    ls.threads = 512;
    ls.lds = (size_t)Rs::NB * Rs::BUF * 16 + (size_t)8 * 16384 + (size_t)T_COUNT * 128 * 4 + 64;
    return sp_launch(k_node_ring_hs, a, ls, s, attr_set);
}
hipError_t launch_project_split_h(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s) {
    static bool attr_set = false;
    return sp_launch(k_project_split_h, a, lc, s, attr_set);
}
hipError_t launch_project_split(const NodeArgs& a, const LaunchCfg& lc, hipStream_t s) {
    static bool attr_set = false;
    return sp_launch(k_project_split, a, lc, s, attr_set);
}

int split_prow_block() { return MGN_PROW_BLOCK; }

}  // namespace mgn
