// NOT BUILT.  k_edge_ring_h with one wave per SIMD (four-wave blocks, 432 registers, no scratch): this tile's e kept in registers (no second
// read), no request inside a chain, the next tile's Q rows requested ahead of this tile's e stores.  Parity green (tests/test_gpu_fp32_split.py
// with MGN_RING_HP=1), and SLOWER: 3.11 ms on M-1M against 2.38 for k_edge_ring_h<8> and 2.74 for k_edge_ring_h<4> (same box).  A single wave
// per SIMD has nobody to hide behind: its ~80 vector-memory instructions per tile cost it 250-500 cycles of issue each while the CU's memory
// pipeline is busy, and the matrix pipe idles through the whole epilogue (docs/experiments.md round 5).  Compiles in csrc/split.hip.
// ================================================================================================
// k_edge_ring_hp: k_edge_ring_h with ONE wave per SIMD (four-wave blocks) and the 512 registers that buys spent on memory: this tile's e
// stays in registers from its arrival to its store (no second read); the next tile's e is requested at this tile's top into a landing
// array of its own, this tile's P rows at its top into the array layer 2 writes later, the next tile's Q rows in the epilogue AHEAD of
// the e stores (which go last): no chain has a request inside it, and nothing a tile waits for first sits behind stores.  Four
// 64-register arrays (e, the two layer arrays, the landing array).
// ================================================================================================
__global__ __launch_bounds__(256, 1) void k_edge_ring_hp(const EdgeArgs a) {
    constexpr int NT = 4, L = 128, PC = 16384, NWV = 4;
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
    u32x4* ringbase = reinterpret_cast<u32x4*>(wl + 3 * PC);
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
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int i = 0; i < W / NWV; ++i) ringbase[w * BUF + i * NWV * 64 + tid] = src.lo[0][w * W * 64 + i * NWV * 64 + tid];
#pragma unroll
        for (int i = 0; i < W / NWV; ++i) {
            pend.v[1][i] = src.lo[(2 * W) / 32][((2 * W) % 32) * 64 + i * NWV * 64 + tid];
            pend.v[0][i] = pend.v[1][i];
        }
    }
    __syncthreads();
    const u32x4* l1h = reinterpret_cast<const u32x4*>(wl);
    const u32x4* l2h = reinterpret_cast<const u32x4*>(wl + PC);
    const u32x4* l3h = reinterpret_cast<const u32x4*>(wl + 2 * PC);
    const float sw1 = a.h2_s[2], rsw1 = a.h2_rs[2], rsw2 = a.h2_rs[0], rsw3 = a.h2_rs[1], b2pos = a.h2_b2pos;
    TileWalk tw0(a.ntiles, 0), tw(a.ntiles, wave);
    if (tw0.tile >= tw0.end) return;
    const int iters = (tw0.end - tw0.tile + tw0.stride - 1) / tw0.stride;
    const int last = a.tile0 + tw0.tile + (iters - 1) * tw0.stride;
    tw.tile += a.tile0;
    tw.end += a.tile0;
    auto clamp = [&](int t) { return t < tw.end ? t : last; };
    f32x16 E[NT], A[NT], B[NT], En[NT];
    EdgeIdx ix = load_edge_idx_nb(a.snd, a.rcv, a.E, clamp(tw.tile), lane0 & 31);
    EdgeIdx ixn = load_edge_idx_nb(a.snd, a.rcv, a.E, clamp(tw.tile + tw.stride), lane0 & 31);
    {
        const int h0 = lane0 >> 5;
        load_frag<NT>(En, tile_ptr(a.Elat, clamp(tw.tile), L, lane0), STRIDE_TILE);
        load_frag<NT>(A, prow_ptr(a.Q, ix.r >= 0 ? ix.r : 0, L, h0), STRIDE_PROW);
    }
    int stamp_tile = 0;
    (void)stamp_tile;
    for (int j = 0; j < iters; ++j, ++stamp_tile) {
        OPAQUE_LANE();
        const bool on = tw.tile < tw.end;
        const int tile = clamp(tw.tile);
        const int nxt = clamp(tw.tile + tw.stride);
        const EdgeIdx ixnn = load_edge_idx_nb(a.snd, a.rcv, a.E, clamp(tw.tile + 2 * tw.stride), c);      // two tiles ahead: the next tile's Q rows are requested in this epilogue
        const bool valid = on && ix.r >= 0;
        const int r = ix.r >= 0 ? ix.r : 0;
        f32x4* etile = tile_ptr(a.Elat, tile, L, lane);
        u32x4* ring = ringbase + lane;
        STAMP(0);
        RhFrag nx = rh_first<W, 0>(l1h, ring, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) E[t] = En[t];
        const H2Scale x1 = h2_scale(h2_rowmax<true>(E));
        {
            const float cinv = x1.s * sw1;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) A[t][k] *= cinv;      // (A holds Q[r]: requested in the epilogue before, ahead of that tile's e stores)
        }
        PHASE_FENCE();
        load_frag<NT>(B, prow_ptr(a.P, ix.s, L, h), STRIDE_PROW);                                         // this tile's P rows (needed behind layer 1) and the
        load_frag<NT>(En, tile_ptr(a.Elat, nxt, L, lane), STRIDE_TILE);                                  // next tile's e (needed at its top): no request inside a chain
        PHASE_FENCE();
        h2_layer_ring<W, 0, 0, 0, NWV>(A, E, l1h, l2h, ring, src, nx, pend, lane, tid, x1.s);            // layer 1: no request inside
        {
            const float c1 = x1.rs * rsw1;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) A[t][k] = __builtin_fmaf(A[t][k], c1, B[t][k]);
        }
        CST(1);
        const H2Scale x2 = h2_scale(h2_rowmax<false>(A));
        zero_frag<NT>(B);
        CST(2);
        h2_layer_ring<W, 1, 1, 0, NWV>(B, A, l2h, l3h, ring, src, nx, pend, lane, tid, x2.s);
        CST(3);
        const float c2 = x2.rs * rsw2;
        const H2Scale x3 = h2_scale(__builtin_fmaf(h2_rowmax<false>(B), c2, b2pos));
        zero_frag<NT>(A);
        CST(4);
        h2_layer_ring<W, 2, 2, 0, NWV>(A, B, l3h, l1h, ring, src, nx, pend, lane, tid, x3.s, c2, tb + T_B2 * L + 4 * h);
        CST(5);
        PHASE_FENCE();
        {   // bias + un-scaling of layer 3, then LayerNorm: A = e'
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
                        const float v = __builtin_fmaf(A[t][4 * g + i], c3, bv[i]);
                        A[t][4 * g + i] = v;
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
                    const float d = A[t][k] - mean;
                    A[t][k] = d;
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
                    for (int i = 0; i < 4; ++i) A[t][4 * g + i] = A[t][4 * g + i] * rstd * gv[i] + bv[i];
                }
            }
        }
        CST(6);
#pragma unroll
        for (int t = 0; t < NT; ++t) E[t] += A[t];                   // e <- e + e'
        CST(7);
        const int reff = ix.r >= 0 ? r : (-4 - c);
        const int rprev = __shfl_up(reff, 1, 32);
        const int rnext = __shfl_down(reff, 1, 32);
        const bool head = (c == 0) || (reff != rprev);
        const unsigned hm = (unsigned)__ballot(head);
        const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
        const int st_in = max(start, c & 16);
        const bool c1 = (c - 1 >= st_in), c2s = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
        const bool cx = (c >= 16) && (start <= 15);
        PHASE_FENCE();
        asm volatile("s_nop 1");
        RG_SCAN_LEVEL(A, c1, "row_shr:1 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(A, c2s, "row_shr:2 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(A, c4, "row_shr:4 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(A, c8, "row_shr:8 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(A, cx, "row_bcast:15 row_mask:0xa bank_mask:0xf");
        PHASE_FENCE();
        const bool tail = valid && ((c == 31) || (reff != rnext));
        const int r_first = __builtin_amdgcn_readfirstlane(reff);
        const bool sl = (start == 0) && (ix.r_before == r_first);
        const bool sr = (c == 31) && (ix.r_after == reff);
        const bool to_carry = sl || sr;
        f32x4* dst = to_carry ? prow_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), L, h) : tile_ptr(a.AGG, r >> 5, L, 32 * h + (r & 31));
        if (tail) store_frag<NT>(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, A);
        PHASE_FENCE();
        load_frag<NT>(A, prow_ptr(a.Q, ixn.r >= 0 ? ixn.r : 0, L, h), STRIDE_PROW);      // the next tile's Q rows AHEAD of this tile's e stores (vmcnt retires in order)
        PHASE_FENCE();
        if (valid) ring_store_e(etile, E);                           // last: nothing the next tile waits for is behind these
        ix = ixn;
        ixn = ixnn;
        tw.tile += tw.stride;
    }
}

