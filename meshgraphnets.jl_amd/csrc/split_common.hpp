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

}  // namespace mgn
