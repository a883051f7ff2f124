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

}  // namespace mgn
