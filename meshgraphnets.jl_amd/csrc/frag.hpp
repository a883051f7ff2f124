// Device-side fragment helpers of the "lane-per-row" MFMA design, shared by the inference
// kernels (kernels.hip) and the training kernels (train.hip).  See kernels.hip's header comment for the design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace mgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define DEVINL __device__ __forceinline__
// Phase fence: values are SSA to the compiler, so "reuse array y for the reload" is only a hint; without a
// fence hipcc hoists the next phase's loads above the current MFMA chain, a fourth 64-VGPR array becomes
// live and the kernel spills.
#define PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
// The two waves that share a SIMD arbitrate instruction issue by priority, then age: without help the older
// wave's MFMA chain starves the younger wave's gather/LayerNorm/scan phase (measured 2-3x longer).  Every
// wave therefore raises its priority while it is OUTSIDE its MFMA chains (-6 % kernel time, same-box A/B).
#ifndef MGN_PRIO
#define MGN_PRIO 1
#endif
// Re-derive the lane id through an opaque asm once per tile: every address and table read that depends
// on it then stays INSIDE the persistent tile loop.  Without this hipcc hoists ~250 loop-invariant LDS
// table reads and 64-bit weight addresses out of the loop and spills them all to scratch.
#define OPAQUE_LANE()                          \
    int lane = lane0;                          \
    asm volatile("" : "+v"(lane));             \
    const int c = lane & 31, h = lane >> 5
constexpr float LN_EPS = 1e-5f;
constexpr int NUM_XCD = 8;

// ------------------------------------------------------------------------------------------------
// fragment <-> memory helpers.  A lane (c,h) owns 4*NT pieces of 4 floats: piece m = 4t+g holds
// features 32t + 8g + 4h + {0..3} of row c and lives in registers x[t][4g..4g+3].  Two memory layouts:
//   row-major   [row][L]            piece m of lane (c,h) at  row*L + (2m+h)*4        -> stride 2 float4
//   tile-major  [tile][m][lane][4]  piece m of lane l     at  tile*32L + m*256 + l*4  -> stride 64 float4
// Tile-major ("fragment-major") arrays make every wave-instruction a contiguous 1 KiB: used for all
// arrays that are only ever touched tile-wise (edge latents, node latents, AGG); arrays gathered by
// index (P, Q, CARRY) stay row-major.  `p` already includes the lane's own offset.
// ------------------------------------------------------------------------------------------------
constexpr int STRIDE_ROW = 2, STRIDE_TILE = 64;

DEVINL const f32x4* row_ptr(const float* base, int64_t row, int L, int h) {
    return reinterpret_cast<const f32x4*>(base + row * L) + h;
}
DEVINL f32x4* row_ptr(float* base, int64_t row, int L, int h) { return reinterpret_cast<f32x4*>(base + row * L) + h; }
// P / Q / CARRY rows (the arrays that are read with lane = row gathers) go through prow_ptr.  MGN_PROW_BLOCK = 8 (the default since
// round 4) stores them in blocks of eight rows, 32-byte piece m of the block's rows side by side -- [row / 8][piece m][row % 8][h][4 floats]
// -- so that a gather instruction (one piece m of 32 rows) finds the pieces of neighbouring rows in the same cache line and the rows of
// a node tile are stored with coalesced 1 KiB instructions (tools/gather_probe.hip: a 32-row gather costs the CU 62 cycles per
// instruction, an 8-line access 16-17): node side - 5 %, step - 1.5 % on M-1M.  It needs a numbering in which the ends of an edge are
// close (on scattered senders a 128-byte line carries 32 useful bytes: - 5..9 % fp32, - 34 % bf16 in round 3) -- which mgn_set_graph
// now guarantees (graph_host.cpp: breadth-first renumbering of a mesh that arrives scattered).  -DMGN_PROW_BLOCK=0: plain row-major rows.
// Piece m of a row: prow_ptr(base, row, L, h)[m * STRIDE_PROW].
#ifndef MGN_PROW_BLOCK
#define MGN_PROW_BLOCK 8
#endif
constexpr int STRIDE_PROW = MGN_PROW_BLOCK ? 2 * MGN_PROW_BLOCK : STRIDE_ROW;
DEVINL int64_t prow_index(int64_t row, int L) {       // f32x4 index of (row, piece 0, h 0)
#if MGN_PROW_BLOCK
    return (row / MGN_PROW_BLOCK) * (int64_t)(MGN_PROW_BLOCK * L / 4) + (row % MGN_PROW_BLOCK) * 2;
#else
    return row * (L / 4);
#endif
}
// f32x4 index of the X-th 16-byte piece of a row counted as in a plain row-major row (X = 2 m + h)
DEVINL int64_t prow_f4(int64_t row, int X, int L) { return prow_index(row, L) + (X >> 1) * STRIDE_PROW + (X & 1); }
DEVINL const f32x4* prow_ptr(const float* base, int64_t row, int L, int h) { return reinterpret_cast<const f32x4*>(base) + prow_index(row, L) + h; }
DEVINL f32x4* prow_ptr(float* base, int64_t row, int L, int h) { return reinterpret_cast<f32x4*>(base) + prow_index(row, L) + h; }
DEVINL const f32x4* tile_ptr(const float* base, int64_t tile, int L, int lane) {
    return reinterpret_cast<const f32x4*>(base + tile * (TILE * L)) + lane;
}
DEVINL f32x4* tile_ptr(float* base, int64_t tile, int L, int lane) {
    return reinterpret_cast<f32x4*>(base + tile * (TILE * L)) + lane;
}

template <int NT>
DEVINL void load_frag(f32x16 (&x)[NT], const f32x4* __restrict__ p, int stride) {
#pragma unroll
    for (int m = 0; m < 4 * NT; ++m) {
        const f32x4 v = p[m * stride];
        x[m >> 2][4 * (m & 3) + 0] = v[0]; x[m >> 2][4 * (m & 3) + 1] = v[1];
        x[m >> 2][4 * (m & 3) + 2] = v[2]; x[m >> 2][4 * (m & 3) + 3] = v[3];
    }
}

template <int NT>
DEVINL void add_frag(f32x16 (&x)[NT], const f32x4* __restrict__ p, int stride) {
#pragma unroll
    for (int m = 0; m < 4 * NT; ++m) {
        const f32x4 v = p[m * stride];
        x[m >> 2][4 * (m & 3) + 0] += v[0]; x[m >> 2][4 * (m & 3) + 1] += v[1];
        x[m >> 2][4 * (m & 3) + 2] += v[2]; x[m >> 2][4 * (m & 3) + 3] += v[3];
    }
}

template <int NT>
DEVINL void store_frag(f32x4* __restrict__ p, int stride, const f32x16 (&x)[NT]) {
#pragma unroll
    for (int m = 0; m < 4 * NT; ++m) {
        f32x4 v;
        v[0] = x[m >> 2][4 * (m & 3) + 0]; v[1] = x[m >> 2][4 * (m & 3) + 1];
        v[2] = x[m >> 2][4 * (m & 3) + 2]; v[3] = x[m >> 2][4 * (m & 3) + 3];
        p[m * stride] = v;
    }
}

// one 16-register quarter (feature block t: pieces 4t..4t+3) of a fragment
DEVINL void load_quarter(f32x16& q, const f32x4* __restrict__ p, int stride, int t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = p[(4 * t + g) * stride];
        q[4 * g + 0] = v[0]; q[4 * g + 1] = v[1]; q[4 * g + 2] = v[2]; q[4 * g + 3] = v[3];
    }
}
DEVINL void store_quarter(f32x4* __restrict__ p, int stride, int t, const f32x16& q) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v;
        v[0] = q[4 * g + 0]; v[1] = q[4 * g + 1]; v[2] = q[4 * g + 2]; v[3] = q[4 * g + 3];
        p[(4 * t + g) * stride] = v;
    }
}

// table (bias / gamma / beta) in fragment order: float4 tab[4*NT][2]
template <int NT>
DEVINL void tab_frag(f32x16 (&x)[NT], const float* tab, int h) {
    const f32x4* t4 = reinterpret_cast<const f32x4*>(tab) + h;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = t4[2 * (4 * t + g)];
            x[t][4 * g + 0] = v[0]; x[t][4 * g + 1] = v[1]; x[t][4 * g + 2] = v[2]; x[t][4 * g + 3] = v[3];
        }
}

template <int NT>
DEVINL void zero_frag(f32x16 (&x)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) x[t][k] = 0.f;
}

// ASM = false: plain fmaxf (two instructions per element, but nothing the scheduler cannot see: k_node_step, which keeps the builtin
// MFMAs, is 7 % slower with the asm form and its wait states)
#ifdef MGN_RELU_FMAXF
constexpr bool RELU_ASM_DEFAULT = false;
#else
constexpr bool RELU_ASM_DEFAULT = true;
#endif
template <int NT, bool ASM = RELU_ASM_DEFAULT>
DEVINL void relu_frag(f32x16 (&x)[NT]) {
    if constexpr (!ASM) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) x[t][k] = fmaxf(x[t][k], 0.f);
        return;
    }
#ifndef MGN_RELU_FMAXF
    // x usually comes straight out of an MFMA chain, and hipcc's hazard recogniser does not cover an inline-asm reader of an MFMA
    // result (16 passes: 18 wait states before a VALU read): the wait states, tied to the registers
    if constexpr (NT == 4) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    else if constexpr (NT == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(x[0]), "+v"(x[1]));
    else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(x[0]));
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#ifdef MGN_RELU_FMAXF
            x[t][k] = fmaxf(x[t][k], 0.f);          // two instructions: hipcc canonicalises the operand first
#else
            float r;                                // one v_max_f32 (maxNum: NaN -> 0, like fmaxf); non-volatile: free to be scheduled
            asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x[t][k]));
            x[t][k] = r;
#endif
        }
#ifndef MGN_RELU_FMAXF
    // ... and the other way round: an inline-asm VALU write followed by a compiler-issued MFMA that reads the register gets no
    // wait states either (k_node_step, builtin MFMAs: wrong results in the streaming and L = 64 / 32 instantiations)
    if constexpr (NT == 4) asm volatile("s_nop 3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    else if constexpr (NT == 2) asm volatile("s_nop 3" : "+v"(x[0]), "+v"(x[1]));
    else asm volatile("s_nop 3" : "+v"(x[0]));
#endif
}

// ------------------------------------------------------------------------------------------------
// One L x L weight chunk:  acc[feature][row] += W^T * in   (16*NT k-steps x NT MFMA 32x32x2 f32)
// Weight fragment layout (host: pack_chunk): w[(j*64 + lane)*NT + t] = W[kbase + phi(j,h)][32t + i],
// lane = 32h + i.  RES: w is in LDS; else global (L2-resident), streamed through a register ring.
// ------------------------------------------------------------------------------------------------
template <int NT> struct AVec;
template <> struct AVec<4> { typedef f32x4 T; };
template <> struct AVec<2> { typedef f32x2 T; };
template <> struct AVec<1> { typedef float T; };

template <int NT> DEVINL float aget(const typename AVec<NT>::T& a, int t) { return a[t]; }
template <> DEVINL float aget<1>(const float& a, int) { return a; }

// tools/issue_probe.hip: independent v_mfma_f32_32x32x2_f32 issued back to back run at 144 TFLOP/s; with >= 16 idle clocks
// (`s_nop 3`) behind each they reach 156.  MGN_MFMA_NOP = N appends `s_nop N` to every MFMA of the chunk chains (non-volatile
// inline asm: hipcc may still move the LDS / global weight reads across it).  The assembler-level MFMA is invisible to hipcc's
// hazard recogniser: MFMA_CHAIN_BEGIN / _END supply the wait states around a chain.
#ifndef MGN_MFMA_NOP
#define MGN_MFMA_NOP 3          // -1: the compiler builtin, no padding (A/B)
#endif
#if MGN_MFMA_NOP >= 0
template <bool PAD>
DEVINL f32x16 mfma32(float a, float b, f32x16 c) {
    if constexpr (PAD) {
        asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_nop %3" : "+v"(c) : "v"(a), "v"(b), "n"(MGN_MFMA_NOP));
        return c;
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
}
// wait states tied to the registers they protect (an untied `s_nop` statement is free to move away from them):
// VALU writes of the operands -> first MFMA; last MFMAs -> VALU reads of the accumulators (16 passes: 18 wait states)
// Safe by construction: the operands of a padded chain may have been written by the VALU (4 wait states would do) OR by a
// compiler-issued MFMA of a neighbouring builtin chain (k_node_step<GEN> mixes the two), whose result a following MFMA may read
// as its B operand only after 18 wait states -- and hipcc's hazard recogniser sees neither side of an inline-asm MFMA.  The
// full 18 are therefore always supplied (14 extra idle clocks per chain of 64 x NT^2 MFMAs).
template <int NT>
DEVINL void mfma_chain_begin(f32x16 (&acc)[NT], const f32x16 (&in)[NT]) {
    if constexpr (NT == 4)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]));
    else if constexpr (NT == 2)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]) : "v"(in[0]), "v"(in[1]));
    else
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0]) : "v"(in[0]));
}
template <int NT>
DEVINL void mfma_chain_end(f32x16 (&acc)[NT]) {
    if constexpr (NT == 4)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    else if constexpr (NT == 2)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
    else
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[0]));
}
#define MFMA_CHAIN_BEGIN(ACC, IN) do { if constexpr (PAD) mfma_chain_begin(ACC, IN); } while (0)
#define MFMA_CHAIN_END(ACC, NT_) do { if constexpr (PAD) mfma_chain_end(ACC); } while (0)
#else
template <bool PAD> DEVINL f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
#define MFMA_CHAIN_BEGIN(ACC, IN) do {} while (0)
#define MFMA_CHAIN_END(ACC, NT_) do {} while (0)
#endif

// PAD = false: the compiler builtin (k_node_step is 6 % slower with the padded form; k_edge_step and k_project are faster)
template <int NT, bool RES, bool PAD = true>
DEVINL void mfma_chunk(f32x16 (&acc)[NT], const f32x16 (&in)[NT], const float* w, int lane) {
    typedef typename AVec<NT>::T AV;
    constexpr int J = 16 * NT;
    const AV* wv = reinterpret_cast<const AV*>(w) + lane;
    MFMA_CHAIN_BEGIN(acc, in);
    if constexpr (RES) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const AV a = wv[j * 64];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = mfma32<PAD>(aget<NT>(a, t), in[j >> 4][j & 15], acc[t]);
        }
    } else {
#ifndef MGN_CHUNK_PF
#define MGN_CHUNK_PF 4
#endif
        constexpr int PF = MGN_CHUNK_PF;  // k-steps in flight: PF x (NT MFMA x 64 cyc) of cover for an L2 hit
        AV ring[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) ring[p] = wv[p * 64];
#ifdef MGN_CHUNK_FENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const AV a = ring[j % PF];
            if (j + PF < J) ring[j % PF] = wv[(j + PF) * 64];
#ifdef MGN_CHUNK_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = mfma32<PAD>(aget<NT>(a, t), in[j >> 4][j & 15], acc[t]);
#ifdef MGN_CHUNK_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }
    MFMA_CHAIN_END(acc, NT);
}

// A chunk whose first JR k-steps are LDS-resident and whose tail streams from L2 (JR = 0: all streamed).  The
// ring is primed before the resident steps, so the first streamed fragments have JR k-steps to arrive.
template <int NT, int JR, bool PAD = true>
DEVINL void mfma_chunk_split(f32x16 (&acc)[NT], const f32x16 (&in)[NT], const float* w_lds, const float* w_glb, int lane) {
    typedef typename AVec<NT>::T AV;
    constexpr int J = 16 * NT;
#ifndef MGN_PF
#define MGN_PF 4
#endif
    constexpr int PF = (J - JR) < MGN_PF ? (J - JR) : MGN_PF;
    const AV* wl = reinterpret_cast<const AV*>(w_lds) + lane;
    const AV* wg = reinterpret_cast<const AV*>(w_glb) + lane;
    AV ring[PF > 0 ? PF : 1];
    MFMA_CHAIN_BEGIN(acc, in);
#pragma unroll
    for (int p = 0; p < PF; ++p) ring[p] = wg[(JR + p) * 64];
#ifdef MGN_SPLIT_FENCE
    __builtin_amdgcn_sched_barrier(0);      // the requests go out HERE, not where hipcc would sink them (just before their use)
#endif
#pragma unroll
    for (int j = 0; j < J; ++j) {
        AV a;
        if (j < JR) {
            a = wl[j * 64];
        } else {
            a = ring[(j - JR) % PF];
            if (j + PF < J) ring[(j - JR) % PF] = wg[(j + PF) * 64];
#ifdef MGN_SPLIT_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = mfma32<PAD>(aget<NT>(a, t), in[j >> 4][j & 15], acc[t]);
#ifdef MGN_SPLIT_FENCE
        if (j >= JR) __builtin_amdgcn_sched_barrier(0);
#endif
    }
    MFMA_CHAIN_END(acc, NT);
}

// LayerNorm variant (spec_variant): slot T_LN of a table block holds (eps_in, eps_out) and
//     rstd = 1 / (sqrt(var + eps_in) + eps_out).
// MGN-spec v1 is (1e-5, 0): (x - mean) / sqrt(var + eps) -- bitwise what 1 / sqrtf(var + eps) gives (adding +0 is exact);
// mgn_config.ln_mode = MGN_LN_STD_EPS makes it (0, 1e-5): (x - mean) / (std + eps), the form some LuxLib 0.5 releases used.
// `gamma`: the block's T_GAMMA slot (L floats per slot).
DEVINL float ln_rstd(float var, const float* gamma, int L) {
    const float* p = gamma + (T_LN - T_GAMMA) * L;
    return 1.0f / (sqrtf(var + p[0]) + p[1]);
}
DEVINL float ln_rstd_at(float var, const float* lnp) { return 1.0f / (sqrtf(var + lnp[0]) + lnp[1]); }   // lnp: the T_LN slot itself

// LayerNorm over the row's L features: 16*NT in this lane + 16*NT in lane^32.  Biased variance.
template <int NT>
DEVINL void layer_norm_frag(f32x16 (&x)[NT], const float* gamma, const float* beta, int h) {
    constexpr float invL = 1.0f / (32 * NT);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) s += x[t][k];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * invL;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float d = x[t][k] - mean;
            x[t][k] = d;
            q += d * d;
        }
    q += __shfl_xor(q, 32, 64);
    const float rstd = ln_rstd(q * invL, gamma, 32 * NT);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma) + h;
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta) + h;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 gv = g4[2 * (4 * t + g)];
            const f32x4 bv = b4[2 * (4 * t + g)];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[t][4 * g + i] = x[t][4 * g + i] * rstd * gv[i] + bv[i];
        }
}

// ------------------------------------------------------------------------------------------------
// cooperative global -> LDS copy of resident weight chunks + tables (once per block; persistent grid)
// ------------------------------------------------------------------------------------------------
// global -> LDS copy of resident weight chunks + tables (once per block; persistent grid).  Deliberately the plain loop: a
// copy with eight loads in flight (copy_to_lds_vec) shortens the prologue from ~20 us to a few, but the persistent fp32
// kernels then run 4 % SLOWER in steady state (same-box A/B, M-1M and a 90 k-node mesh) -- once more the alternation of the
// two waves of a SIMD, which the staggered end of the slow prologue happens to help.
DEVINL void copy_to_lds(float* dst, const float* __restrict__ src, int nfloats) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    for (int i = threadIdx.x; i < nfloats / 4; i += blockDim.x) d4[i] = s4[i];
}

// Eight 16-byte loads of a thread in flight before the first LDS write (a plain `d4[i] = s4[i]` loop waits for every load
// before issuing the next one, ~1 us per iteration).  Used by the bf16 kernels, whose 96 KiB preload is paid per launch
// and dominated a cylinder-sized mesh (M-flag, two edge sets: 132 -> 78 us per step).
DEVINL void copy_to_lds_vec(f32x4* d4, const f32x4* __restrict__ s4, int n4) {
    constexpr int B = 8;
    const int bd = blockDim.x;
    for (int i = threadIdx.x; i < n4; i += B * bd) {
        f32x4 v[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int k = i + u * bd;
            v[u] = s4[k < n4 ? k : i];
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int k = i + u * bd;
            if (k < n4) d4[k] = v[u];
        }
    }
}

// weight preload of a persistent launch: the plain loop for long launches, eight loads in flight when the launch has only a few
// tiles per wave (the ~20 us prologue of the plain loop is then a large share of the kernel: node side of a 125 k-node mesh)
DEVINL void copy_to_lds_sel(float* dst, const float* __restrict__ src, int nfloats, bool fast) {
    if (fast) copy_to_lds_vec(reinterpret_cast<f32x4*>(dst), reinterpret_cast<const f32x4*>(src), nfloats / 4);
    else copy_to_lds(dst, src, nfloats);
}

// ------------------------------------------------------------------------------------------------
// Cooperative 4-wave tiles (L = 128): wave t owns feature block t of every layer's output; see kernels.hip
// ("Cooperative-tile kernels for SMALL graphs") for the design.  Shared by kernels.hip and train.hip.
// ------------------------------------------------------------------------------------------------
#ifndef MGN_COOP_PF
#define MGN_COOP_PF 4
#endif
constexpr int COOP_PF = MGN_COOP_PF;   // weight ring depth in 16-byte fragments (4 k-steps each)

// wt: this wave's t-slice of a chunk in t-major order [j/4][lane][4].  The weight ring of a chain can be primed ahead of
// time (coop_prime) -- before the previous chain or the exchange barrier -- so that the first fragments' L2 latency
// (~1.5 k cycles per chain on a small mesh, where nothing else hides it) is off the critical path.
struct CoopRing {
    f32x4 r[COOP_PF];
};
DEVINL void coop_prime(CoopRing& ring, const float* wt, int lane) {
    const f32x4* wv = reinterpret_cast<const f32x4*>(wt) + lane;
#pragma unroll
    for (int p = 0; p < COOP_PF; ++p) ring.r[p] = wv[p * 64];
}
// FENCE pins the ring with scheduling fences: group m consumes slot m % COOP_PF and requests group m + COOP_PF before its
// four MFMAs.  Left alone hipcc sinks every request to just before its use (an effective depth of 1-2), which costs a launch
// with one or two tiles per CU 2-3 % (nothing else hides the L2 round trips) -- but the fences take away the freedom the
// scheduler needs when several tiles per SIMD alternate (+4 % at 6 tiles per CU, +11 % at 12): the launch wrappers pick.
template <bool FENCE>
DEVINL void coop_chain_primed(f32x16& acc, const f32x16 (&in)[4], const float* wt, int lane, CoopRing& ring) {
    const f32x4* wv = reinterpret_cast<const f32x4*>(wt) + lane;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const f32x4 a = ring.r[m % COOP_PF];
        if (m + COOP_PF < 16) ring.r[m % COOP_PF] = wv[(m + COOP_PF) * 64];
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], in[m >> 2][4 * (m & 3) + i], acc, 0, 0, 0);
        if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
}
template <bool FENCE>
DEVINL void coop_chain(f32x16& acc, const f32x16 (&in)[4], const float* wt, int lane) {
    CoopRing ring;
    coop_prime(ring, wt, lane);
    if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
    coop_chain_primed<FENCE>(acc, in, wt, lane, ring);
}

// every wave publishes its 16-register slice and reads back the full 64-register row fragment
DEVINL void coop_exchange(f32x16 (&full)[4], const f32x16& mine, f32x4* xch, int wave, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v;
        v[0] = mine[4 * g + 0]; v[1] = mine[4 * g + 1]; v[2] = mine[4 * g + 2]; v[3] = mine[4 * g + 3];
        xch[(4 * wave + g) * 64 + lane] = v;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const f32x4 v = xch[m * 64 + lane];
        full[m >> 2][4 * (m & 3) + 0] = v[0]; full[m >> 2][4 * (m & 3) + 1] = v[1];
        full[m >> 2][4 * (m & 3) + 2] = v[2]; full[m >> 2][4 * (m & 3) + 3] = v[3];
    }
}

DEVINL void tab_quarter(f32x16& q, const float* tab, int t, int h) { load_quarter(q, reinterpret_cast<const f32x4*>(tab) + h, 2, t); }

DEVINL void relu_quarter(f32x16& q) {
#pragma unroll
    for (int k = 0; k < 16; ++k) q[k] = fmaxf(q[k], 0.f);
}

// LayerNorm statistics from the full row fragment, applied to this wave's quarter
DEVINL void coop_layer_norm(f32x16& mine, const f32x16 (&full)[4], const float* gamma, const float* beta, int t, int h) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) s += full[u][k];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / 128);
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float d = full[u][k] - mean;
            q += d * d;
        }
    q += __shfl_xor(q, 32, 64);
    const float rstd = ln_rstd(q * (1.0f / 128), gamma, 128);
    f32x16 gq, bq;
    tab_quarter(gq, gamma, t, h);
    tab_quarter(bq, beta, t, h);
#pragma unroll
    for (int k = 0; k < 16; ++k) mine[k] = (mine[k] - mean) * rstd * gq[k] + bq[k];
}

// same with gamma / beta quarters already in registers (lnp: the block's T_LN slot)
DEVINL void coop_layer_norm_reg(f32x16& mine, const f32x16 (&full)[4], const f32x16& gq, const f32x16& bq, const float* lnp) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) s += full[u][k];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / 128);
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float d = full[u][k] - mean;
            q += d * d;
        }
    q += __shfl_xor(q, 32, 64);
    const float rstd = ln_rstd_at(q * (1.0f / 128), lnp);
#pragma unroll
    for (int k = 0; k < 16; ++k) mine[k] = (mine[k] - mean) * rstd * gq[k] + bq[k];
}

// XCD-aware persistent tile walk: blocks b and b+8 share an XCD (round-robin dispatch), so every XCD
// gets one contiguous range of tiles and its waves sweep it interleaved -> gathered P/Q rows of
// neighbouring tiles are served by that XCD's L2.  Speed only; any placement is correct.
#ifndef MGN_SPREAD_ROUNDS
#define MGN_SPREAD_ROUNDS 24
#endif
#ifndef MGN_SPREAD_ROUNDS_NODE
#define MGN_SPREAD_ROUNDS_NODE MGN_SPREAD_ROUNDS
#endif
struct TileWalk {
    int tile, end, stride;
    DEVINL TileWalk(int ntiles, int wave, int spread_rounds = MGN_SPREAD_ROUNDS) {
        const int wpb = blockDim.x >> 6;
        const int xcd = blockIdx.x % NUM_XCD;
        const int per = (ntiles + NUM_XCD - 1) / NUM_XCD;
        const int nb = (gridDim.x - xcd + NUM_XCD - 1) / NUM_XCD;  // blocks on this XCD label
        end = min(xcd * per + per, ntiles);
        stride = nb * wpb;
        // Few rounds (mid-size meshes, per-GPU partitions): the last, partly filled round must not land on the first blocks of
        // the XCD alone -- with positions numbered block-major, 2 813 node tiles gave 12 CUs per XCD 16 tiles and the other
        // 20 CUs 8.  Wave-major positions hand one extra tile to wave 0 (1, 2 ..) of EVERY block instead: 11 per CU.
        const bool spread = per < spread_rounds * stride;
        tile = xcd * per + (spread ? wave * nb + (int)(blockIdx.x / NUM_XCD) : (int)(blockIdx.x / NUM_XCD) * wpb + wave);
    }
};


}  // namespace mgn
