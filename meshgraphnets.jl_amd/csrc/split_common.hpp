// Helpers of the fp32-accurate "exact split" kernels (split.hip, split_ws.hip): an fp32 value as three bf16 pieces.
#pragma once
#include "tile_common.hpp"

namespace mgn {

typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

DEVINL float sp_f(unsigned u) { return __builtin_bit_cast(float, u); }
DEVINL unsigned sp_u(float f) { return __builtin_bit_cast(unsigned, f); }
#ifndef MGN_SP_CVT_ASM
#define MGN_SP_CVT_ASM 1
#endif
DEVINL unsigned sp_cvt_pk(float a, float b) {            // v_cvt_pk_bf16_f32: round to nearest even
#if MGN_SP_CVT_ASM
    // as one opaque instruction: from the cast form hipcc converts the low element a second time (alone) for its unpacked copy
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    sp_bf16x2 p;
    p[0] = (__bf16)a;
    p[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, p);
#endif
}
// ReLU on the bits: a signed-integer max with 0 (negative floats, -0 and negative NaNs -> +0): one instruction the compiler can see
DEVINL float sp_relu(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// three-way split of a pair of values: v = hi + mid + lo exactly, each piece a bf16 (11 VALU instructions, 13 with the ReLU)
struct SpPieces {            // the hi / mid / lo pieces of one k-step of a lane: 8 bf16 each (4 dwords)
    unsigned h[4], m[4], l[4];
};
DEVINL sp_bf16x8 sp_op(const unsigned (&v)[4]) {
    u32x4 q;
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
    return __builtin_bit_cast(sp_bf16x8, q);
}
template <bool RELU>
DEVINL void sp_split_pair(unsigned& hi, unsigned& mid, unsigned& lo, float v0, float v1) {
    if constexpr (RELU) {
        v0 = sp_relu(v0);
        v1 = sp_relu(v1);
    }
    const unsigned h = sp_cvt_pk(v0, v1);
    const float r0 = v0 - sp_f(h << 16), r1 = v1 - sp_f(h & 0xffff0000u);
    const unsigned m = sp_cvt_pk(r0, r1);
    const float q0 = r0 - sp_f(m << 16), q1 = r1 - sp_f(m & 0xffff0000u);
    hi = h;
    mid = m;
    lo = sp_cvt_pk(q0, q1);
}
DEVINL sp_bf16x8 sp_wop(const u32x4& v) { return __builtin_bit_cast(sp_bf16x8, v); }

// ---- two fp16 pieces (round 5; split.hip: k_edge_ring_h and the node-side kernels behind MGN_FP32_SPLIT = 1) -----------------------------
// An fp32 value times a power of two s is hi + lo with hi, lo fp16 (11 + 11 significand bits and the sign of lo: 23 of the 24 bits;
// |x s - hi - lo| <= 2^-23 |x s|, zero three times out of four), as long as lo is a NORMAL fp16 -- so every operand is scaled first:
// weights per L x L chunk on the host (its largest entry lands in [2^14, 2^15)), activations per ROW in the kernel (lane = row: the
// scale is one register, the output column of a row is un-scaled by the same power of two, exactly).  Of the four piece products three
// are kept (lo x lo <= 2^-22 relative, random sign: against float64 a layer measures 0.9e-7 with three and with four products; six bf16
// products 0.4e-7; a plain fp32 GEMM 4.6e-7 -- tools/f16_split_accuracy.py).  v_mfma_f32_32x32x16_f16 multiplies exactly, accumulates in fp32 and
// keeps subnormal inputs (tools/f16_probe.hip), and a value can never overflow: the scale is taken from the row's own maximum.
typedef _Float16 sp_f16x8 __attribute__((ext_vector_type(8)));
DEVINL sp_f16x8 h2_op(const unsigned (&v)[4]) {
    u32x4 q;
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
    return __builtin_bit_cast(sp_f16x8, q);
}
DEVINL sp_f16x8 h2_wop(const u32x4& v) { return __builtin_bit_cast(sp_f16x8, v); }
constexpr unsigned H2_EXP_MIN = 87u << 23;       // rows / chunks whose largest magnitude is below 2^-40 are scaled as if it were 2^-40
struct H2Scale { float s, rs; };                 // s = 2^(14 - floor(log2(amax))), rs = 1 / s
DEVINL H2Scale h2_scale(float amax) {
    unsigned eb = sp_u(amax) & 0x7f800000u;
    eb = eb > H2_EXP_MIN ? eb : H2_EXP_MIN;
    return {sp_f((268u << 23) - eb), sp_f(eb - (14u << 23))};
}
// the pieces of a pair of values scaled by s.  FIN 0: the values as they are; 1: ReLU first; 2: ReLU(v c + b) first (the bias and the
// un-scaling of the layer before, folded into the split that consumes its accumulators).  7 / 9 / 11 VALU instructions, no packed ones.
template <int FIN>
DEVINL void h2_split_pair(unsigned& hi, unsigned& lo, float v0, float v1, float s, float c = 0.f, float b0 = 0.f, float b1 = 0.f) {
    if constexpr (FIN == 2) {
        v0 = __builtin_fmaf(v0, c, b0);
        v1 = __builtin_fmaf(v1, c, b1);
    }
    if constexpr (FIN >= 1) {
        v0 = sp_relu(v0);
        v1 = sp_relu(v1);
    }
    unsigned h, l;
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(v0), "v"(s));            // RN16(v0 s) -> low half
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(v1), "v"(s));            // RN16(v1 s) -> high half
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v0), "v"(s), "v"(h));   // v0 s - hi: exact
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v1), "v"(s), "v"(h));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(r0), "v"(r1));
    hi = h;
    lo = l;
}
// largest |x| (ABS) or largest x, at least 0 (!ABS: what survives a ReLU) of a lane's 64 values and of its row's other half
template <bool ABS>
DEVINL float h2_rowmax(const f32x16 (&x)[4]) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            const float a = ABS ? __builtin_fabsf(x[t][k]) : x[t][k], b = ABS ? __builtin_fabsf(x[t][k + 1]) : x[t][k + 1];
            m = __builtin_fmaxf(m, __builtin_fmaxf(a, b));
        }
    return __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
}

// Buffer descriptors (a wave-uniform 64-bit base in four scalar registers) + a 32-bit byte offset per lane + a scalar / immediate offset:
// one address register per stream where 64-bit pointers cost a pair per 4 KiB of reach.  Used where registers are the limit.
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
struct N16Buf { __amdgpu_buffer_rsrc_t r; };
DEVINL N16Buf n16_buf(const void* base, int bytes = -1) { return {__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000)}; }
constexpr unsigned N16_DROP = 0x80000000u;      // a lane offset beyond every descriptor's range: the store is dropped, without a branch
DEVINL f32x4 n16_ld(const N16Buf& b, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)voff, soff, 0));
}
DEVINL void n16_st(const N16Buf& b, unsigned voff, int soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), b.r, (int)voff, soff, 0);
}
template <int AUX>      // cache policy bits of the buffer instruction: 1 sc0, 2 nt, 16 sc1
DEVINL f32x4 n16_ld_pol(const N16Buf& b, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)voff, soff, AUX));
}
template <int AUX>
DEVINL void n16_st_pol(const N16Buf& b, unsigned voff, int soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), b.r, (int)voff, soff, AUX);
}
DEVINL u32x4 n16_ldu(const N16Buf& b, unsigned voff, int soff) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)voff, soff, 0)); }


#ifndef MGN_SP2_INTERLEAVE
#define MGN_SP2_INTERLEAVE 1      // 1: pin "one MFMA, n VALU" groups inside every (s, t) step (sched_group_barrier)
#endif
// One L x L layer on two fp16 pieces, `in` split on the fly with the row scale sx (split.hip: the node-side kernels; train.hip: the
// streaming training kernels).  p_hi LDS-resident; p_lo LDS-resident too (GL = false) or streamed from L2 through ONE per-wave register
// ring D fragments deep that can be carried from chain to chain (carry / PRIMED / NEXT / nx_lo: see sp_layer_otf in split.hip).
// Refill (RFN > 0; round 6): the registers of `in`'s k-step s are dead once step s has begun (its pieces were computed during step s - 1), so
// k-steps 0 .. RFN - 1 are refilled, as they are released, with pieces 2 s, 2 s + 1 of what the caller needs NEXT from memory (rf: the
// lane's pointer to piece 0, RFS: f32x4 stride between pieces) -- the loads travel while the chain runs; pieces 2 RFN .. 15 are the caller's
// (h2_load_tail).  `in` is written through the const reference in that case only.
template <int D> struct SpRingH { u32x4 r[D]; };
// RFBUF: the refill's requests as buffer loads (rfb: the tile's descriptor, lane offset 16 lane, piece m at m KiB; tile-major arrays only).
template <bool GL, int FIN, int D, int OFF = 0, bool PRIMED = false, bool NEXT = false, int RFN = 0, int RFS = 0, bool RFBUF = false>
DEVINL void h2_layer_otf(f32x16 (&acc)[4], const f32x16 (&in)[4], const u32x4* p_hi, const u32x4* p_lo, int lane, float sx, float cfin = 0.f,
                         const float* btab = nullptr, SpRingH<D>* carry = nullptr, const u32x4* nx_lo = nullptr, const f32x4* rf = nullptr,
                         const N16Buf* rfb = nullptr) {
    f32x16 (&inw)[4] = const_cast<f32x16 (&)[4]>(in);
    const u32x4* w1 = p_hi + lane;
    const u32x4* w2 = p_lo + lane;
    const N16Buf b2 = n16_buf(GL ? p_lo : nullptr);
    const unsigned voff = (unsigned)lane * 16u;
    static_assert(!(PRIMED || NEXT) || GL, "the ring carry-over belongs to a streamed lo piece");
    u32x4 r2[GL ? D : 1];
    if constexpr (GL) {
        if constexpr (PRIMED) {
#pragma unroll
            for (int d = 0; d < D; ++d) r2[d] = carry->r[d];
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) r2[(d + OFF) % D] = n16_ldu(b2, voff, d * 1024);
        }
    }
    const N16Buf c2 = n16_buf(NEXT ? nx_lo : nullptr);
    auto bias = [&](int sn, int u) {
        f32x2 b = {0.f, 0.f};
        if constexpr (FIN == 2) b = *reinterpret_cast<const f32x2*>(btab + 8 * (4 * (sn >> 1) + 2 * (sn & 1) + (u >> 1)) + 2 * (u & 1));
        return b;
    };
    u32x4 n1 = w1[0], n2;
    if constexpr (!GL) n2 = w2[0];
    unsigned ph[4], pl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const f32x2 b = bias(0, u);
        h2_split_pair<FIN>(ph[u], pl[u], in[0][2 * u], in[0][2 * u + 1], sx, cfin, b[0], b[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        unsigned nh[4], nl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int it = 4 * s + t;
            const u32x4 a1 = n1;
            u32x4 a2;
            if constexpr (GL) a2 = r2[(it + OFF) % D]; else a2 = n2;
            if (it + 1 < 32) {
                n1 = w1[(it + 1) * 64];
                if constexpr (!GL) n2 = w2[(it + 1) * 64];
            }
            if constexpr (GL) {
                if (it + D < 32) r2[(it + OFF) % D] = n16_ldu(b2, voff, (it + D) * 1024);
                else if constexpr (NEXT) r2[(it + OFF) % D] = n16_ldu(c2, voff, (it + D - 32) * 1024);
            }
            if constexpr (RFN > 0) {
                if ((t & 1) && s < RFN) {                              // registers of k-step s, half t >> 1
                    f32x4 v;
                    if constexpr (RFBUF) v = n16_ld(*rfb, (unsigned)lane * 16u, (2 * s + (t >> 1)) * 1024);
                    else v = rf[(2 * s + (t >> 1)) * RFS];
#pragma unroll
                    for (int i = 0; i < 4; ++i) inw[s >> 1][8 * (s & 1) + 4 * (t >> 1) + i] = v[i];
                }
            }
            if (s < 7) {
                const int sn = s + 1;
                const f32x2 b = bias(sn, t);
                h2_split_pair<FIN>(nh[t], nl[t], in[sn >> 1][8 * (sn & 1) + 2 * t], in[sn >> 1][8 * (sn & 1) + 2 * t + 1], sx, cfin, b[0], b[1]);
            }
            const sp_f16x8 bh = h2_op(ph), bl = h2_op(pl);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a2), bh, acc[t], 0, 0, 0);      // small terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a1), bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a1), bh, acc[t], 0, 0, 0);
#if MGN_SP2_INTERLEAVE
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ph[u] = nh[u];
            pl[u] = nl[u];
        }
    }
    if constexpr (NEXT) {
#pragma unroll
        for (int d = 0; d < D; ++d) carry->r[d] = r2[d];
    }
}
// pieces 2 RFN .. 15 of a fragment (what a chain's refill leaves to its caller)
template <int RFN, int RFS>
DEVINL void h2_load_tail(f32x16 (&x)[4], const f32x4* p) {
#pragma unroll
    for (int m = 2 * RFN; m < 16; ++m) {
        const f32x4 v = p[m * RFS];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[m >> 2][4 * (m & 3) + i] = v[i];
    }
}
// pieces M0 .. 15 of a tile-major fragment through the tile's descriptor
template <int M0>
DEVINL void h2_load_tile_buf(f32x16 (&x)[4], const N16Buf& b, int lane) {
#pragma unroll
    for (int m = M0; m < 16; ++m) {
        const f32x4 v = n16_ld(b, (unsigned)lane * 16u, m * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[m >> 2][4 * (m & 3) + i] = v[i];
    }
}
template <int NT>
DEVINL void h2_scale_frag(f32x16 (&x)[NT], float c) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) x[t][k] *= c;
}
// x <- x c + table (fragment order)
template <int NT>
DEVINL void h2_finish_frag(f32x16 (&x)[NT], float c, const float* tab, int h) {
    const f32x4* t4 = reinterpret_cast<const f32x4*>(tab) + h;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = t4[2 * (4 * t + g)];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[t][4 * g + i] = __builtin_fmaf(x[t][4 * g + i], c, bv[i]);
        }
}

// ---- cooperative 4-wave tiles (train.hip: k_mlp_fwd_coop / k_mlp_bwd_coop): wave t computes feature block t of acc += in W from the FULL
// input row it holds, so the row scale is the wave's own business.  wp: the chunk's pieces + t * 64 + lane (hi fragment of k-step s at
// wp[256 s], lo at wp[2048 + 256 s]); the sixteen fragments of a chain go through a ring four deep that can be primed ahead of time.
constexpr int H2C_PF = 4;
struct H2CoopRing { u32x4 r[H2C_PF]; };
DEVINL const u32x4* h2c_w(const float* chunk, int t, int lane) { return reinterpret_cast<const u32x4*>(chunk + 2 * 128 * 128) + t * 64 + lane; }
DEVINL u32x4 h2c_frag(const u32x4* wp, int m) { return wp[(m & 1) * 2048 + (m >> 1) * 256]; }      // fragment m of the chain: hi, lo of k-step m / 2
DEVINL void h2c_prime(H2CoopRing& g, const u32x4* wp) {
#pragma unroll
    for (int m = 0; m < H2C_PF; ++m) g.r[m] = h2c_frag(wp, m);
}
DEVINL void h2c_chain_primed(f32x16& acc, const f32x16 (&in)[4], const u32x4* wp, H2CoopRing& g, float rsw) {
    const H2Scale sx = h2_scale(h2_rowmax<true>(in));
    f32x16 part;
#pragma unroll
    for (int k = 0; k < 16; ++k) part[k] = 0.f;
    unsigned ph[4], pl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) h2_split_pair<0>(ph[u], pl[u], in[0][2 * u], in[0][2 * u + 1], sx.s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const u32x4 a1 = g.r[(2 * s) % H2C_PF], a2 = g.r[(2 * s + 1) % H2C_PF];
        if (2 * s + H2C_PF < 16) {
            g.r[(2 * s) % H2C_PF] = h2c_frag(wp, 2 * s + H2C_PF);
            g.r[(2 * s + 1) % H2C_PF] = h2c_frag(wp, 2 * s + 1 + H2C_PF);
        }
        unsigned nh[4], nl[4];
        if (s < 7) {
            const int sn = s + 1;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                h2_split_pair<0>(nh[u], nl[u], in[sn >> 1][8 * (sn & 1) + 2 * u], in[sn >> 1][8 * (sn & 1) + 2 * u + 1], sx.s);
        }
        const sp_f16x8 bh = h2_op(ph), bl = h2_op(pl);
        part = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a2), bh, part, 0, 0, 0);      // small terms first
        part = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a1), bl, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2_wop(a1), bh, part, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ph[u] = nh[u];
            pl[u] = nl[u];
        }
    }
    const float c = sx.rs * rsw;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = __builtin_fmaf(part[k], c, acc[k]);
}

// out = in W (true units), nothing live across the chain but the input
DEVINL void h2_chunk_set(f32x16 (&out)[4], const f32x16 (&in)[4], const float* w, int lane, float rsw) {
    const H2Scale sx = h2_scale(h2_rowmax<true>(in));
    zero_frag<4>(out);
    const u32x4* p = reinterpret_cast<const u32x4*>(w);
    h2_layer_otf<false, 0, 1>(out, in, p, p + 2048, lane, sx.s);
    const float c = sx.rs * rsw;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) out[t][k] *= c;
}
// acc += in W without a third array: the accumulator is put INTO the chain's units (a power of two: exact while |acc| x scale < 2^127,
// i.e. |acc| < 2^19 in the worst case of a 2^-40 row against a 2^-40 chunk), the chain accumulates on top, and it is brought back
DEVINL void h2_chunk_inplace(f32x16 (&acc)[4], const f32x16 (&in)[4], const float* w, int lane, float rsw) {
    const H2Scale sx = h2_scale(h2_rowmax<true>(in));
    const float up = sx.s * (1.0f / rsw), c = sx.rs * rsw;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] *= up;
    const u32x4* p = reinterpret_cast<const u32x4*>(w);
    h2_layer_otf<false, 0, 1>(acc, in, p, p + 2048, lane, sx.s);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] *= c;
}
// acc += in W for an fp32 accumulator in TRUE units: the chunk's two fp16 pieces (hi at w, lo 2048 fragments behind; multiplied by
// 1 / rsw when they were packed) in LDS, the row's scale taken here, the partial sum un-scaled as it is added.  `part`: 64 registers
// of scratch.  What the streaming training kernels call where they called mfma_chunk.
DEVINL void h2_chunk(f32x16 (&acc)[4], const f32x16 (&in)[4], f32x16 (&part)[4], const float* w, int lane, float rsw) {
    const H2Scale sx = h2_scale(h2_rowmax<true>(in));
    zero_frag<4>(part);
    const u32x4* p = reinterpret_cast<const u32x4*>(w);
    h2_layer_otf<false, 0, 1>(part, in, p, p + 2048, lane, sx.s);
    const float c = sx.rs * rsw;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = __builtin_fmaf(part[t][k], c, acc[t][k]);
}

}  // namespace mgn
