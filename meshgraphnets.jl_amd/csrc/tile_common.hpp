// Device helpers shared by the tile kernels of kernels.hip and split.hip: start stagger, diagnostic stamps, the segmented
// DPP scan of the scatter-add, edge index loads, the aggregate load of the node kernels.  See kernels.hip for the design.
#pragma once
#include "frag.hpp"

namespace mgn {

#ifndef MGN_FAST_PRELOAD_TILES
#define MGN_FAST_PRELOAD_TILES 8192     // node-side launches of up to 4 tiles per wave copy their weights with eight loads in flight
#endif

// Waves w and w+4 of a block share a SIMD and run the same program; started together they stay in
// lockstep (both gather, then both want the MFMA pipe).  Delaying the second half once by about half a
// tile period makes one wave's memory/VALU phase coincide with its partner's MFMA chain.
DEVINL void stagger_second_half(int wave, int units) {
    if (wave >= 4)
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(127);
}

// Diagnostic build only (-DMGN_DIAG_STAMPS): s_memtime stamps per tile phase; no stamp executes in the
// shipped kernel.  Values go to a buffer no other code reads.
#ifdef MGN_DIAG_STAMPS
#define STAMP(slot)                                                                              \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long _t;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (a.stamps && blockIdx.x < 4 && stamp_tile < 24 && lane0 == 0)                         \
            a.stamps[(((size_t)blockIdx.x * 8 + wave) * 24 + stamp_tile) * 8 + (slot)] = _t;     \
    } while (0)
// 16-row cooperative kernels: one half tile per block, 1024 blocks x 4 waves x 8 slots
#define STAMP16(slot)                                                                            \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long _t;                                                                   \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (a.stamps && blockIdx.x < 1024 && lane0 == 0)                                         \
            a.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + (slot)] = _t;                         \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#define STAMP16(slot) do {} while (0)
#endif

template <int CTRL, int ROWMASK>
DEVINL float dpp_zero(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, true));
}

// Segmented inclusive sum over runs of equal receiver inside a 32-row tile (lanes c = 0..31 of each half wave), in place on
// NG groups of 16 registers.  c1 .. c8: "my run reaches back at least 1 / 2 / 4 / 8 rows inside my 16-lane DPP row",
// cx: "my run continues from the previous DPP row".  One v_fmac_f32_dpp per register and level:
//     v += dpp(v) * (condition ? 1 : 0)          (a single rounding: bitwise the sum v + dpp(v))
// instead of v_add_f32_dpp + v_cndmask: 320 fewer VALU instructions per fp32 edge tile, and VALU issue time adds to MFMA time
// on this part: edge kernel -2.4 % on M-1M.  DPP reads run with every lane active (a masked-out source
// lane would read as 0 on gfx9), hence the multiplicative mask; invalid source lanes (row boundaries) read 0 (bound_ctrl).
// Inline asm: hipcc has no builtin that yields the DPP form of fmac.  Hazards: a DPP read needs 2 wait states after a VALU
// write of the same VGPR -- consecutive levels touch a register NG * 16 >= 16 instructions apart, and an s_nop covers the
// first level.  MGN_SCAN_FMAC = 0 restores the two-instruction form (also the reference for tests of this helper).
#ifndef MGN_SCAN_FMAC
#define MGN_SCAN_FMAC 1
#endif
// SKIP8: the reach-8 level runs only when some run of the tile reaches back 8 rows (in-degree >= 9; a wave-uniform branch).
// bf16 edge kernel 0.918 -> 0.907 ms on M-1M; the fp32 kernel gets 3 % SLOWER with the branch (same-box A/B), so only bf16 uses it.
template <int NG, bool SKIP8 = false>
DEVINL void segmented_scan(f32x16 (&acc)[NG], bool c1, bool c2, bool c4, bool c8, bool cx) {
#if MGN_SCAN_FMAC
    const float m1 = c1 ? 1.f : 0.f, m2 = c2 ? 1.f : 0.f, m4 = c4 ? 1.f : 0.f, m8 = c8 ? 1.f : 0.f, mx = cx ? 1.f : 0.f;
#define MGN_SCAN_LEVEL(M, CTRL)                                                                                  \
    _Pragma("unroll") for (int t = 0; t < NG; ++t)                                                               \
        _Pragma("unroll") for (int k = 0; k < 16; ++k)                                                           \
            asm volatile("v_fmac_f32_dpp %0, %0, %1 " CTRL " bound_ctrl:0" : "+v"(acc[t][k]) : "v"(M));
    PHASE_FENCE();
    asm volatile("s_nop 1");
    MGN_SCAN_LEVEL(m1, "row_shr:1 row_mask:0xf bank_mask:0xf")
    MGN_SCAN_LEVEL(m2, "row_shr:2 row_mask:0xf bank_mask:0xf")
    MGN_SCAN_LEVEL(m4, "row_shr:4 row_mask:0xf bank_mask:0xf")
    if (!SKIP8 || __builtin_amdgcn_ballot_w64(c8) != 0) {
        MGN_SCAN_LEVEL(m8, "row_shr:8 row_mask:0xf bank_mask:0xf")
    }
    MGN_SCAN_LEVEL(mx, "row_bcast:15 row_mask:0xa bank_mask:0xf")
#undef MGN_SCAN_LEVEL
    PHASE_FENCE();
#else
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        PHASE_FENCE();   // bound the scan's temporaries to one 16-register group at a time
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float v = acc[t][k];
            float u;
            u = v + dpp_zero<0x111, 0xF>(v); v = c1 ? u : v;   // row_shr:1
            u = v + dpp_zero<0x112, 0xF>(v); v = c2 ? u : v;   // row_shr:2
            u = v + dpp_zero<0x114, 0xF>(v); v = c4 ? u : v;   // row_shr:4
            u = v + dpp_zero<0x118, 0xF>(v); v = c8 ? u : v;   // row_shr:8
            u = v + dpp_zero<0x142, 0xA>(v); v = cx ? u : v;   // row_bcast:15 into rows 1 and 3
            acc[t][k] = v;
        }
    }
#endif
}


struct EdgeIdx {
    int s, r, r_before, r_after;
};

DEVINL EdgeIdx load_edge_idx(const EdgeArgs& a, int tile, int c) {
    EdgeIdx ix;
    const int64_t e0 = (int64_t)tile * TILE;
    const int64_t eid = e0 + c;
    const bool valid = eid < a.E;
    ix.s = valid ? a.snd[eid] : 0;
    ix.r = valid ? a.rcv[eid] : -1;                                   // -1 marks a padding lane
    ix.r_before = (tile > 0) ? a.rcv[e0 - 1] : -2;                    // wave-uniform
    ix.r_after = (e0 + TILE < a.E) ? a.rcv[e0 + TILE] : -3;           // wave-uniform
    return ix;
}

// Branch-free form for the software-pipelined kernels: every load is unconditional (addresses clamped into the arrays), the
// padding / boundary cases are selects on the loaded values.  With the conditional form hipcc puts `valid ? snd[eid] : 0` into an
// EXEC-masked block together with everything that depends on it (the 64-bit row address of the later gather), and that block
// starts with s_waitcnt vmcnt(0): the index load requested "ahead of time" is waited for on the spot, with every prefetch in
// flight behind it.  Needs E >= 1 (a launch has at least one tile).
DEVINL EdgeIdx load_edge_idx_nb(const int32_t* __restrict__ snd, const int32_t* __restrict__ rcv, int64_t E, int tile, int c) {
    EdgeIdx ix;
    const int64_t e0 = (int64_t)tile * TILE;
    const int64_t eid = e0 + c;
    const bool valid = eid < E;
    const int64_t ec = valid ? eid : E - 1;
    const int s_ = snd[ec], r_ = rcv[ec];
    const int rb = rcv[e0 > 0 ? e0 - 1 : 0];                           // wave-uniform
    const int ra = rcv[e0 + TILE < E ? e0 + TILE : E - 1];             // wave-uniform
    ix.s = valid ? s_ : 0;
    ix.r = valid ? r_ : -1;                                            // -1 marks a padding lane
    ix.r_before = (tile > 0) ? rb : -2;
    ix.r_after = (e0 + TILE < E) ? ra : -3;
    return ix;
}


// `fast`: launches of a few tiles per wave, where the preload is a large share of the kernel (see copy_to_lds_vec); long
// persistent launches keep the plain loop (the batched copy costs them ~1 % in steady state, same-box A/B on M-1M)
DEVINL void copy_to_lds16(uint16_t* dst, const uint16_t* __restrict__ src, int n, bool fast) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    if (fast) copy_to_lds_vec(d4, s4, n / 8);
    else
        for (int i = threadIdx.x; i < n / 8; i += blockDim.x) d4[i] = s4[i];
}

// PROJECT: also emit P,Q of the next step in the same launch (fewer launches: used for small meshes; on large
// meshes the projection runs as k_project, where both of its chunks are LDS-resident).
// aggregated messages of the tile's nodes: the node's AGG slot, or carry rows when its edge run straddles edge tiles.
// A macro on purpose: as a (force-inlined) function the same code costs k_node_step<4,*> 37 spilled VGPRs.
// Uses tile, nn, valid, lane, h, L of the enclosing tile loop.
#ifndef MGN_AGG_PEEL
#define MGN_AGG_PEEL 1
#endif
#define LOAD_AGGREGATE(NT_, y_, rowptr_, AGG_, CARRY_, zero_row_)                                                        \
    do {                                                                                                                 \
        const int a0 = valid ? (rowptr_)[nn] : 0, a1 = valid ? (rowptr_)[nn + 1] : 0;                                     \
        const int T1 = a0 >> 5, T2 = (a1 - 1) >> 5;                                                                      \
        const int extra = (a1 > a0 && T2 > T1) ? (T2 - T1) : 0;                                                          \
        const bool from_agg = (a1 > a0) && !extra;                                                                       \
        const f32x4* src0 = from_agg ? tile_ptr((AGG_), tile, L, lane)                                                   \
                                     : prow_ptr((CARRY_), extra ? (int64_t)(2 * T1 + 1) : (zero_row_), L, h);            \
        load_frag<NT_>(y_, src0, from_agg ? STRIDE_TILE : STRIDE_PROW);                                                   \
        /* a run that straddles ONE tile boundary (every 32 edges one node's does) is the common case: its second carry row   \
           without a loop and without a branch (the other lanes add the zero row); only hub nodes enter the loop behind it -- \
           as the loop's first trip this cost every tile 64 loop-carried register copies and a spill that was reloaded, with  \
           s_waitcnt vmcnt(0), in the middle of the next chain */                                                            \
        if (MGN_AGG_PEEL) {                                                                                              \
            if (__any(extra >= 1))                                                                                       \
                add_frag<NT_>(y_, prow_ptr((CARRY_), extra >= 1 ? (int64_t)2 * (T1 + 1) : (zero_row_), L, h), STRIDE_PROW); \
            if (__any(extra >= 2))                                                                                       \
                for (int q = 2; __any(q <= extra); ++q)                                                                  \
                    if (q <= extra) add_frag<NT_>(y_, prow_ptr((CARRY_), (int64_t)2 * (T1 + q), L, h), STRIDE_PROW);      \
        } else {                                                                                                         \
            for (int q = 1; __any(q <= extra); ++q)                                                                      \
                if (q <= extra) add_frag<NT_>(y_, prow_ptr((CARRY_), (int64_t)2 * (T1 + q), L, h), STRIDE_PROW);          \
        }                                                                                                                \
    } while (0)

}  // namespace mgn
