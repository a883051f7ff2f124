// NOT BUILT.  Round-2 experiment, kept for the record (docs/experiments.md, "software-pipelined fp32 edge step").
// One wave per SIMD (512 registers); the epilogue of tile i-1 (LayerNorm, residual, segmented scan) is sliced over the k-steps of
// the three MFMA chains of tile i, operands of tile i+1 are requested at the chain breaks.  Bitwise equal to k_edge_step<4, 2>
// (tools/sp_check.py on a 170 x 170 mesh) but SLOWER on M-1M: 4.99 ms against 4.78 ms per launch, same box.
//   * with EMPTY hooks the kernel takes 4.67 ms: a lone wave per SIMD does not reach the 49 k cycles of its 768 MFMAs per tile
//     either (LDS weight-fragment latency behind the per-k-step fences, three serial chain breaks of ~420 instructions);
//   * the hooks then add their full issue time (+0.32 ms): hipcc places them behind the four MFMAs of a k-step, where one 48-cycle
//     gap is left, instead of between them;
//   * asking for the spread (sched_group_barrier per k-step), a deeper LDS fragment ring, or ReLU / bias slices in the hooks each
//     tipped the register allocator over (VGPR / AGPR split 208-240 / 96-128 with 800-1100 bytes of scratch in the loop).
// Fragment of csrc/kernels.hip as it stood (uses frag.hpp / kernels.h helpers: mfma fragments, load_edge_idx_nb, dpp_zero, TileWalk).
// ================================================================================================
// Software-pipelined fp32 edge step: ONE wave per SIMD (four waves per block, up to 512 registers per wave).
// Why: on this part VALU issue time ADDS to MFMA time inside a wave unless the VALU instructions sit in the shadow of an MFMA (a
// v_mfma_f32_32x32x2_f32 occupies the matrix pipe for 64 cycles and the SIMD's vector issue for 16 of them), and two symmetric waves
// per SIMD (k_edge_step) overlap one wave's epilogue with the partner's chain only when their phases happen to differ (MFMA pipe
// busy 80 %).  Here ONE instruction stream carries both: every k-step of the three MFMA chains of tile i (4 MFMAs = 256 cycles of
// matrix pipe, 64 of vector issue) is followed by a SLICE of the epilogue of tile i-1 -- LayerNorm, residual, segmented scan -- and
// of the register hand-over to tile i+1, all independent of the running chain; requests for tile i+1 go out at the chain breaks.
//   chain 1 (e tile x W1e on top of P[s] + Q[r])   k-steps  0..15  row sums        16..31 centred squares    32..63 scale + residual
//   break 1    ReLU, bias;  store e of tile i-1;  request the e tile of i+1
//   chain 2                                         k-step j: all five scan levels of register j of e' (registers are independent)
//   break 2    ReLU, bias;  aggregate stores of tile i-1;  request P[s], Q[r] of tile i+1
//   chain 3                                         k-steps 32..63: P[s] + Q[r] of tile i+1
// Same arithmetic in the same order per element as k_edge_step<4, 2>: bitwise the same results.
// ================================================================================================
template <int CTRL, int ROWMASK>
DEVINL float scan_step(float v, float m) { return __builtin_fmaf(dpp_zero<CTRL, ROWMASK>(v), m, v); }   // v += dpp(v) * m

// k-step loops of one L x L chunk (NT = 4) with a per-k-step hook; hook(j) runs after the four MFMAs of k-step j were issued
// Order request for the current k-step's scheduling region: the hook's VALU instructions go BETWEEN the four MFMAs (each MFMA
// holds the vector issue for 16 of its 64 cycles: ~12 single-issue slots per MFMA), not behind the last one, where only one such
// gap is left before the next k-step's MFMAs are due (measured: behind the MFMAs the hooks cost their full issue time).
#ifndef SP_VALU_PER_MFMA
#define SP_VALU_PER_MFMA 8
#endif
DEVINL void sp_spread() {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, SP_VALU_PER_MFMA, 0);    // then up to SP_VALU_PER_MFMA VALU instructions
    }
}

// (the fragment of k-step j + SP_LDS_PF is requested before the MFMAs of k-step j: behind the per-k-step fence a read issued at
// the top of its own k-step arrives ~130 cycles later, after the matrix pipe has drained the previous four MFMAs)
#ifndef SP_LDS_PF
#define SP_LDS_PF 1
#endif
template <int JR, class Hook>
DEVINL void sp_chunk_split(f32x16 (&acc)[4], const f32x16 (&in)[4], const float* w_lds, const float* w_glb, int lane, Hook&& hook) {
    constexpr int J = 64;
    constexpr int PF = (J - JR) < MGN_PF ? (J - JR) : MGN_PF;
    constexpr int D = SP_LDS_PF;
    const f32x4* wl = reinterpret_cast<const f32x4*>(w_lds) + lane;
    const f32x4* wg = reinterpret_cast<const f32x4*>(w_glb) + lane;
    f32x4 ring[PF > 0 ? PF : 1], lring[D];
#pragma unroll
    for (int p = 0; p < PF; ++p) ring[p] = wg[(JR + p) * 64];
#pragma unroll
    for (int p = 0; p < D; ++p) lring[p] = wl[p * 64];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        f32x4 w;
        if (j < JR) {
            w = lring[j % D];
            if (j + D < JR) lring[j % D] = wl[(j + D) * 64];
        } else {
            w = ring[(j - JR) % PF];
            if (j + PF < J) ring[(j - JR) % PF] = wg[(j + PF) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t], in[j >> 4][j & 15], acc[t], 0, 0, 0);
#ifndef SP_NOHOOKS
        hook(j);
#endif
        sp_spread();
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <class Hook>
DEVINL void sp_chunk_lds(f32x16 (&acc)[4], const f32x16 (&in)[4], const float* w_lds, int lane, Hook&& hook) {
    constexpr int D = SP_LDS_PF;
    const f32x4* wl = reinterpret_cast<const f32x4*>(w_lds) + lane;
    f32x4 lring[D];
#pragma unroll
    for (int p = 0; p < D; ++p) lring[p] = wl[p * 64];
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const f32x4 w = lring[j % D];
        if (j + D < 64) lring[j % D] = wl[(j + D) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t], in[j >> 4][j & 15], acc[t], 0, 0, 0);
#ifndef SP_NOHOOKS
        hook(j);
#endif
        sp_spread();
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(256, 1) void k_edge_step_sp(const EdgeArgs a) {
    constexpr int NT = 4, L = 128, CH = 16 * NT * 64 * NT;
    constexpr int JR = MGN_EDGE_JR, PART = JR * 64 * NT;
    constexpr float invL = 1.0f / L;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    copy_to_lds(smem, a.chunk[0], CH);
    copy_to_lds(smem + CH, a.chunk[1], CH);
    copy_to_lds(smem + 2 * CH, a.chunk[2], PART);
    float* tb = smem + 2 * CH + PART;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* w2 = smem;
    const float* w3 = smem + CH;
    TileWalk tw(a.ntiles, wave);
    tw.tile += a.tile0;
    tw.end += a.tile0;
    if (tw.tile >= tw.end) return;
    const int last = tw.tile + ((tw.end - 1 - tw.tile) / tw.stride) * tw.stride;
    auto clampt = [&](int t) { return t <= last ? t : last; };
    f32x16 x[NT], acc[NT], y[NT], e1[NT], xp[NT];
    EdgeIdx ix = load_edge_idx_nb(a.snd, a.rcv, a.E, tw.tile, lane0 & 31);
    EdgeIdx ixn = load_edge_idx_nb(a.snd, a.rcv, a.E, clampt(tw.tile + tw.stride), lane0 & 31);
    EdgeIdx ixp = ix;
    {
        const int h0 = lane0 >> 5;
        load_frag<NT>(acc, row_ptr(a.P, ix.s, L, h0), STRIDE_ROW);
        add_frag<NT>(acc, row_ptr(a.Q, ix.r >= 0 ? ix.r : 0, L, h0), STRIDE_ROW);
        load_frag<NT>(x, tile_ptr(a.Elat, tw.tile, L, lane0), STRIDE_TILE);
        zero_frag<NT>(e1);
    }
    bool have_prev = false;
    int tile_prev = tw.tile;
    // scan structure of the PREVIOUS tile (both halves see the same structure); a macro: stays in the caller's scope
#define SP_SCAN_SETUP()                                                                                             \
    const bool validp = ixp.r >= 0;                                                                                 \
    const int rp_ = validp ? ixp.r : 0;                                                                             \
    const int reff = validp ? rp_ : (-4 - c);                                                                       \
    const int rprev = __shfl_up(reff, 1, 32);                                                                       \
    const int rnext = __shfl_down(reff, 1, 32);                                                                     \
    const bool head = (c == 0) || (reff != rprev);                                                                  \
    const unsigned hm = (unsigned)__ballot(head);                                                                   \
    const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));                                            \
    const int st_in = max(start, c & 16);                                                                           \
    const float m1 = (c - 1 >= st_in) ? 1.f : 0.f, m2 = (c - 2 >= st_in) ? 1.f : 0.f, m4 = (c - 4 >= st_in) ? 1.f : 0.f,            \
                m8 = (c - 8 >= st_in) ? 1.f : 0.f, mx = ((c >= 16) && (start <= 15)) ? 1.f : 0.f;                   \
    const bool sp_tail = validp && ((c == 31) || (reff != rnext));                                                  \
    const int r_first = __builtin_amdgcn_readfirstlane(reff);                                                       \
    const bool sl = (start == 0) && (ixp.r_before == r_first);                                                      \
    const bool sr = (c == 31) && (ixp.r_after == reff);                                                             \
    const bool sp_carry = sl || sr;                                                                                 \
    f32x4* sp_dst = sp_carry ? row_ptr(a.CARRY, (int64_t)2 * tile_prev + (sl ? 0 : 1), L, h)                        \
                             : tile_ptr(a.AGG, rp_ >> 5, L, 32 * h + (rp_ & 31));
#define SP_SCAN_REG(v)                                                                                              \
    do {                                                                                                            \
        v = scan_step<0x111, 0xF>(v, m1);                                                                           \
        v = scan_step<0x112, 0xF>(v, m2);                                                                           \
        v = scan_step<0x114, 0xF>(v, m4);                                                                           \
        v = scan_step<0x118, 0xF>(v, m8);                                                                           \
        v = scan_step<0x142, 0xA>(v, mx);                                                                           \
    } while (0)
#define SP_REG(arr, r) arr[(r) >> 4][(r) & 15]
    for (;;) {
        OPAQUE_LANE();
        const int tile = tw.tile;
        const int next = tile + tw.stride;
        const bool has_next = next <= last;
        const EdgeIdx ixnn = load_edge_idx_nb(a.snd, a.rcv, a.E, clampt(tile + 2 * tw.stride), c);
        const f32x4* g4 = reinterpret_cast<const f32x4*>(tb + T_GAMMA * L) + h;
        const f32x4* b4 = reinterpret_cast<const f32x4*>(tb + T_BETA * L) + h;
        // the previous tile's e rows AGAIN, for its residual (32 k-steps = ~8 k cycles ahead of their use): keeping them in
        // registers from their first use made 6 x 64 live registers in chain 3 and spilled; the re-read costs HBM bandwidth this
        // MFMA-bound kernel has to spare
        load_frag<NT>(xp, tile_ptr(a.Elat, tile_prev, L, lane), STRIDE_TILE);
        // ---- chain 1; its k-steps carry the LayerNorm + residual of the previous tile (layer_norm_frag's sums in
        //      layer_norm_frag's order)
        float sum = 0.f, sq = 0.f, mean = 0.f, rstd = 0.f;
        sp_chunk_split<JR>(acc, x, smem + 2 * CH, a.chunk[2], lane, [&](int j) {
            if (j < 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) sum += SP_REG(e1, 4 * j + u);
                if (j == 15) {
                    sum += __shfl_xor(sum, 32, 64);
                    mean = sum * invL;
                }
            } else if (j < 32) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 4 * (j - 16) + u;
                    const float d = SP_REG(e1, k) - mean;
                    SP_REG(e1, k) = d;
                    sq += d * d;
                }
                if (j == 31) {
                    sq += __shfl_xor(sq, 32, 64);
                    rstd = 1.0f / sqrtf(sq * invL + LN_EPS);
                }
            } else if ((j & 1) == 0) {          // one 4-register piece (one gamma / beta float4) every other k-step
                const int m = (j - 32) >> 1, t = m >> 2, g = m & 3;
                const f32x4 gv = g4[2 * m], bv = b4[2 * m];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = e1[t][4 * g + i] * rstd * gv[i] + bv[i];
                    e1[t][4 * g + i] = e;                                   // e' of the previous tile
                    xp[t][4 * g + i] += e;                                  // e <- e + e'
                }
            }
        });
        // ---- break 1
        relu_frag<NT>(acc);
        tab_frag<NT>(y, tb + T_B2 * L, h);
        if (have_prev && ixp.r >= 0) store_frag<NT>(tile_ptr(a.Elat, tile_prev, L, lane), STRIDE_TILE, xp);
        load_frag<NT>(x, tile_ptr(a.Elat, clampt(next), L, lane), STRIDE_TILE);    // e tile of the next tile (chain 1 was x's last use)
        SP_SCAN_SETUP()
        __builtin_amdgcn_sched_barrier(0);
        // ---- chain 2, k-step j: the five scan levels of register j of e' (registers are independent)
        sp_chunk_lds(y, acc, w2, lane, [&](int j) { SP_SCAN_REG(SP_REG(e1, j)); });
        // ---- break 2
        relu_frag<NT>(y);
        tab_frag<NT>(acc, tb + T_B3 * L, h);
        if (have_prev && sp_tail) store_frag<NT>(sp_dst, sp_carry ? STRIDE_ROW : STRIDE_TILE, e1);
        load_frag<NT>(e1, row_ptr(a.P, ixn.s, L, h), STRIDE_ROW);          // next tile: P[s] ...
        load_frag<NT>(xp, row_ptr(a.Q, ixn.r >= 0 ? ixn.r : 0, L, h), STRIDE_ROW);   // ... and Q[r]
        __builtin_amdgcn_sched_barrier(0);
        // ---- chain 3, second half: P[s] + Q[r] of the next tile, two registers per k-step
        sp_chunk_lds(acc, y, w3, lane, [&](int j) {
            if (j >= 32) {
                SP_REG(e1, 2 * (j - 32)) += SP_REG(xp, 2 * (j - 32));
                SP_REG(e1, 2 * (j - 32) + 1) += SP_REG(xp, 2 * (j - 32) + 1);
            }
        });
#pragma unroll
        for (int t = 0; t < NT; ++t) {          // hand-over: e1 <-> acc
            const f32x16 pq = e1[t];
            e1[t] = acc[t];                      // pre-LayerNorm output of this tile
            acc[t] = pq;                         // P[s] + Q[r] of the next tile
        }
        ixp = ix;
        ix = ixn;
        ixn = ixnn;
        tile_prev = tile;
        have_prev = true;
        if (!has_next) break;
        tw.tile = next;
    }
    // ---- drain: the epilogue of the wave's last tile
    {
        OPAQUE_LANE();
        load_frag<NT>(xp, tile_ptr(a.Elat, tile_prev, L, lane), STRIDE_TILE);
        layer_norm_frag<NT>(e1, tb + T_GAMMA * L, tb + T_BETA * L, h);
#pragma unroll
        for (int t = 0; t < NT; ++t) xp[t] += e1[t];
        if (ixp.r >= 0) store_frag<NT>(tile_ptr(a.Elat, tile_prev, L, lane), STRIDE_TILE, xp);
        SP_SCAN_SETUP()
#pragma unroll
        for (int r = 0; r < 64; ++r) SP_SCAN_REG(SP_REG(e1, r));
        if (sp_tail) store_frag<NT>(sp_dst, sp_carry ? STRIDE_ROW : STRIDE_TILE, e1);
    }
#undef SP_SCAN_SETUP
#undef SP_SCAN_REG
#undef SP_REG
}

