// NOT BUILT.  The weight-gradient kernel on two fp16 pieces (round 5): parity green (tests/test_gpu_training_*.py ran on it), and no
// faster -- 30.0 vs 29.0 us per launch on the cylinder mesh (the launch is bound by the latency of a block's row loop and by the 60 MB of
// per-block partials it writes), 0.474 vs 0.461 s per mgn_step on M-1M (HBM-bound on the row reads).  It compiles in csrc/train.hip beside
// k_wgrad (split_common.hpp helpers); launch it where launch_wgrad launches k_wgrad<4>.
// The same on two fp16 pieces (L = 128; round 5): the reduction dimension is the ROW index, so an operand's scale has to be the same for
// all rows of a k-step -- one power of two per wave for X and one for G, taken from a RUNNING maximum over the block's rows (the maximum
// of the first step to begin with; when a later step exceeds it -- a wave-uniform branch, a handful of times per block -- the
// accumulators are brought down by the same power of two).  v_mfma_f32_32x32x16_f16: A = X^T (lane (m, kh): feature 32 ti + m, rows
// 8 kh .. 8 kh + 7 of the 16-row step), B = G (lane (n, kh): feature 32 t + n, the same rows); 12 MFMAs of 32 cycles per 16 rows
// instead of 32 of 64.  Against the float64 oracle the gradients stay inside the tolerances of the fp32 kernels (the sum over thousands
// of rows dominates both).
DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__global__ __launch_bounds__(256) void k_wgrad_h(const WgradBatch wb) {
    constexpr int NT = 4, L = 128;
    const WgradJob& jb = wb.job[blockIdx.y];
    const int64_t r0 = (int64_t)blockIdx.x * wb.rows_per_block;
    if (r0 >= jb.rows) return;
    const int64_t r1 = r0 + wb.rows_per_block < jb.rows ? r0 + wb.rows_per_block : jb.rows;
    const int lane = threadIdx.x & 63, m = lane & 31, kh = lane >> 5;
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
    unsigned ea = 0, eb = 0;                     // exponent fields of the running maxima of |X| and |G| (wave-uniform; 0: none yet)
    for (int64_t q = r0; q < r1; q += 16) {
        float av[8], bv[NT][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t row = q + 8 * kh + u;
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
                bv[t][u] = ok ? gb : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) bs[t] += bv[t][u];
        if (!with_w) continue;
        float ma = 0.f, mb = 0.f;
#pragma unroll
        for (int u = 0; u < 8; u += 2) ma = __builtin_fmaxf(ma, __builtin_fmaxf(__builtin_fabsf(av[u]), __builtin_fabsf(av[u + 1])));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < 8; u += 2) mb = __builtin_fmaxf(mb, __builtin_fmaxf(__builtin_fabsf(bv[t][u]), __builtin_fabsf(bv[t][u + 1])));
        unsigned na = __builtin_amdgcn_readfirstlane(sp_u(wave_max(ma)) & 0x7f800000u);
        unsigned nb = __builtin_amdgcn_readfirstlane(sp_u(wave_max(mb)) & 0x7f800000u);
        na = na > H2_EXP_MIN ? na : H2_EXP_MIN;
        nb = nb > H2_EXP_MIN ? nb : H2_EXP_MIN;
        if (na > ea || nb > eb) {                // a larger operand than any before: the units of the accumulators follow (first step: from nothing)
            const unsigned ta = na > ea ? na : ea, tb_ = nb > eb ? nb : eb;
            if (ea != 0) {
                const float down = sp_f((127u << 23) - (ta - ea) - (tb_ - eb));      // 2^-(exponent steps of X + of G)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[t][k] *= down;
            }
            ea = ta;
            eb = tb_;
        }
        const float sa = sp_f((268u << 23) - ea), sb = sp_f((268u << 23) - eb);      // (h2_scale's s for the running maxima)
        unsigned ah[4], al[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) h2_split_pair<0>(ah[u], al[u], av[2 * u], av[2 * u + 1], sa);
        const sp_f16x8 Ah = h2_op(ah), Al = h2_op(al);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            unsigned bh[4], bl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) h2_split_pair<0>(bh[u], bl[u], bv[t][2 * u], bv[t][2 * u + 1], sb);
            const sp_f16x8 Bh = h2_op(bh), Bl = h2_op(bl);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, acc[t], 0, 0, 0);
        }
    }
    // D layout of the 32x32 MFMA: register r of lane l holds D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]
    if (with_w) {
        const float c = ea ? sp_f(ea - (14u << 23)) * sp_f(eb - (14u << 23)) : 0.f;      // 1 / (sa sb)
        float* pw = jb.pw + (size_t)blockIdx.x * L * L;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) pw[(size_t)(32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kh) * L + 32 * t + m] = acc[t][r] * c;
    }
    if (ti == 0 && jb.pb) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float sb_ = bs[t] + __shfl_xor(bs[t], 32, 64);
            if (kh == 0) jb.pb[(size_t)blockIdx.x * L + 32 * t + m] = sb_;
        }
    }
}

