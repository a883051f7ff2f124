// NOT BUILT.  Three opt-in kernels of the split path (three bf16 pieces, six products) as they stood at the end of round 4, removed
// from csrc/split.hip in round 5: parity-green, measured, never the default (docs/experiments.md rounds 3-4):
//   k_edge_split2 (MGN_FP32_SPLIT=2)  per-wave register rings for the streamed weight pieces: 160 KiB of L2 traffic per tile and wave
//   k_node_ring   (MGN_NODE_RING=1)   node MLP + projection in one lock-step launch over an LDS ring: 1.215 vs 1.173 ms
//   k_edge_ring2  (MGN_FP32_SPLIT=3)  two independent four-wave blocks per CU: 4.0-4.1 vs 3.4 ms
// They use the helpers of csrc/split.hip (sp_layer_otf, Rg, ring_barrier, RG_SCAN_LEVEL, ...) and compile in its place.
// ================================================================================================
// Processor edge step (K3 + K4 + K5) on the split path.  chunk order as in k_edge_step: split[0] = W2, [1] = W3, [2] = W1[2L:3L].
// LDS: hi of W1e, hi + mid of W2, hi of W3 (128 KiB) + tables.  The e tile is re-read for the residual (its registers carry the
// second layer's output meanwhile).
// ================================================================================================
__global__ __launch_bounds__(512, 2) void k_edge_split2(const EdgeArgs a) {
    constexpr int NT = 4, L = 128, PC = 16384;                      // PC: bf16 elements per piece
    constexpr int D = MGN_SP2_D, D1 = MGN_SP2_D1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    {
        const bool fast = a.ntiles <= 16 * 1024;
        copy_to_lds16(wl, a.split[2], PC, fast);
        copy_to_lds16(wl + PC, a.split[0], 2 * PC, fast);           // hi + mid of W2 are adjacent
        copy_to_lds16(wl + 3 * PC, a.split[1], PC, fast);
    }
    float* tb = smem + 4 * PC / 2;
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    __syncthreads();
    const u32x4* l1h = reinterpret_cast<const u32x4*>(wl);
    const u32x4* l2h = reinterpret_cast<const u32x4*>(wl + PC);
    const u32x4* l2m = reinterpret_cast<const u32x4*>(wl + 2 * PC);
    const u32x4* l3h = reinterpret_cast<const u32x4*>(wl + 3 * PC);
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.split[2]);      // 2048 fragments per piece
    const u32x4* g2 = reinterpret_cast<const u32x4*>(a.split[0]);
    const u32x4* g3 = reinterpret_cast<const u32x4*>(a.split[1]);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stagger_second_half(wave, a.stagger);
    TileWalk tw(a.ntiles, wave);
    tw.tile += a.tile0;
    tw.end += a.tile0;
    if (tw.tile >= tw.end) return;
    f32x16 acc[NT], y[NT];
    EdgeIdx ix = load_edge_idx_nb(a.snd, a.rcv, a.E, tw.tile, lane0 & 31);
    {   // first tile: layer-1 accumulator P[s] + Q[r] (Q carries b1) and the e tile
        const int h0 = lane0 >> 5;
        load_frag<NT>(acc, prow_ptr(a.P, ix.s, L, h0), STRIDE_PROW);
        add_frag<NT>(acc, prow_ptr(a.Q, ix.r >= 0 ? ix.r : 0, L, h0), STRIDE_PROW);
        load_frag<NT>(y, tile_ptr(a.Elat, tw.tile, L, lane0), STRIDE_TILE);
    }
    int stamp_tile = 0;
    (void)stamp_tile;
    for (;; ++stamp_tile) {
        OPAQUE_LANE();
        const int tile = tw.tile;
        const int next = tile + tw.stride;
        const bool has_next = next < tw.end;
        const int nxt = has_next ? next : tile;
        const EdgeIdx ixn = load_edge_idx_nb(a.snd, a.rcv, a.E, nxt, c);
        const bool valid = ix.r >= 0;
        const int r = valid ? ix.r : 0;
        f32x4* etile = tile_ptr(a.Elat, tile, L, lane);
        STAMP(0);
        __builtin_amdgcn_s_setprio(0);
        sp_layer_otf<true, true, false, D>(acc, y, l1h, g1 + 2048, g1 + 4096, lane);   // layer 1 (edge part); y = e tile
        STAMP(1);
        tab_frag<NT>(y, tb + T_B2 * L, h);
        STAMP(2);
        sp_layer_otf<false, true, true, D1>(y, acc, l2h, l2m, g2 + 4096, lane);        // layer 2 (ReLU folded into the split)
        STAMP(3);
        tab_frag<NT>(acc, tb + T_B3 * L, h);
        STAMP(4);
        sp_layer_otf<true, true, true, D>(acc, y, l3h, g3 + 2048, g3 + 4096, lane);    // layer 3
        STAMP(5);
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_PRIO);                        // memory / VALU phase: win issue arbitration
        load_frag<NT>(y, etile, STRIDE_TILE);                        // e again, for the residual
        layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);   // acc = e'
        STAMP(6);
#pragma unroll
        for (int t = 0; t < NT; ++t) y[t] += acc[t];                 // e <- e + e'
        if (valid) store_frag<NT>(etile, STRIDE_TILE, y);            // padding rows of the last tile stay zero
        STAMP(7);
        // ---- segmented sum of e' over runs of equal receiver: as in k_edge_step
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
        const bool tail = valid && ((c == 31) || (reff != rnext));
        const int r_first = __builtin_amdgcn_readfirstlane(reff);
        const bool sl = (start == 0) && (ix.r_before == r_first);
        const bool sr = (c == 31) && (ix.r_after == reff);
        const bool to_carry = sl || sr;
        f32x4* dst = to_carry ? prow_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), L, h) : tile_ptr(a.AGG, r >> 5, L, 32 * h + (r & 31));
        if (tail) store_frag<NT>(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, acc);
        if (!has_next) break;
        PHASE_FENCE();
        // turnover: the next tile's layer-1 accumulator and e tile
        load_frag<NT>(acc, prow_ptr(a.P, ixn.s, L, h), STRIDE_PROW);
        add_frag<NT>(acc, prow_ptr(a.Q, ixn.r >= 0 ? ixn.r : 0, L, h), STRIDE_PROW);
        load_frag<NT>(y, tile_ptr(a.Elat, nxt, L, lane), STRIDE_TILE);
        ix = ixn;
        tw.tile = next;
    }
}


// ================================================================================================
// Node MLP AND the P / Q projection of the next step in one launch, lock-step like k_edge_ring: the six chunks (W1v, W1a, W2, W3, WP,
// WQ; 18 pieces, 576 KiB) pass through ONE LDS ring per block -- three window buffers of 24 KiB (8 (s, t) steps of hi, mid and lo),
// 24 windows per tile, every thread fetches 3 x 16 bytes per window.  k_node_split + k_project_split stream 320 KiB of pieces per tile
// and WAVE from L2 (what-if: 0.25 ms of the node side's 1.19), read V' again for the projection and need two launches; here a tile
// costs 72 KiB of L2 traffic per wave and v' goes from the residual straight into the projection.  For a step that is followed by
// another one on the same handle (mgn_proc_node(project_next)), one edge set, one partition.
// OPT-IN (MGN_NODE_RING=1): parity-green, and no faster -- 1.215 vs 1.173 ms on M-1M.  The chains run at pipe speed here too; what
// the node side loses is outside them (138 k cycles per round against 84 k of chains): the aggregate rows behind two dependent
// row-pointer loads, the V tile read again, and the P / Q rows written lane = row (32 cache lines per store instruction).
// ================================================================================================
template <int W>
struct Rn {
    static constexpr int WPL = 32 / W;          // windows per chunk
    static constexpr int BUF = 3 * W * 64;      // u32x4 elements per window buffer: [piece (hi, mid, lo)][step][lane]
};
struct RnFrag {
    u32x4 h, m, l;
};
template <int W>
DEVINL RnFrag rn_read(const u32x4* ring /* + lane */, int gw, int step) {
    RnFrag f;
    const u32x4* b = ring + (gw % 3) * Rn<W>::BUF + step * 64;
    f.h = b[0];
    f.m = b[W * 64];
    f.l = b[2 * W * 64];
    return f;
}
// chunk CH of NCH (NCH * WPL = 0 mod 3: a window's buffer is the same for every tile); src[c]: the three pieces of chunk c, 2048
// fragments apart.  nx: the fragments of step 0 in, those of the next chunk's step 0 out.
// RF: `in` is refilled, as the split releases its registers, with the sixteen 16-byte pieces rf[m * rfstride] (sp_layer_ring's
// refill, schedule 2: one request every second step, pieces of k-step s + 2 into the registers of k-step s; the caller requested
// pieces 0 .. 3 into `side` before the layer); `in` comes back holding them in fragment order.
template <int W, int CH, int NCH, bool RELU, int NWV = 8, bool RF = false>
DEVINL void spn_layer(f32x16 (&acc)[4], f32x16 (&in)[4], u32x4* ring, const u32x4* const (&src)[NCH], RnFrag& nx, int lane, int tid,
                      const f32x4* rf = nullptr, int rfstride = 0, const f32x4* side = nullptr) {
    constexpr int WPL = Rn<W>::WPL, NW = NCH * WPL, BUF = Rn<W>::BUF;
    constexpr int LPT = 8 / NWV;                 // fragments per piece and thread in a window
    static_assert(NW % 3 == 0, "window -> buffer must not depend on the tile");
    SpPieces p;
#pragma unroll
    for (int u = 0; u < 4; ++u) sp_split_pair<RELU>(p.h[u], p.m[u], p.l[u], in[0][2 * u], in[0][2 * u + 1]);
    u32x4 ld[3 * LPT];                           // this thread's share of window gw + 2 on its way to LDS
    unsigned voff = (unsigned)tid * 16u;
    asm volatile("" : "+v"(voff));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        SpPieces n;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int it = 4 * s + t;
            const int gw = WPL * CH + it / W;
            const u32x4 a1 = nx.h, a2 = nx.m, a3 = nx.l;
            if (it % W == 0) {                                         // request window gw + 2
                const int g2 = (gw + 2) % NW, c2 = g2 / WPL, w2 = g2 % WPL;
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < LPT; ++i)
                        ld[q * LPT + i] =
                            *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src[c2] + q * 2048 + w2 * W * 64 + i * NWV * 64) + voff);
            }
            if constexpr (RF) {
                if ((t & 1) && s < 6) {                                // registers of k-step s (free since the step began), half t >> 1
                    const f32x4 v = rf[(2 * (s + 2) + (t >> 1)) * rfstride];
#pragma unroll
                    for (int i = 0; i < 4; ++i) in[s >> 1][8 * (s & 1) + 4 * (t >> 1) + i] = v[i];
                }
            }
            if (it + 1 < 32) nx = rn_read<W>(ring, WPL * CH + (it + 1) / W, (it + 1) % W);
            else nx = rn_read<W>(ring, (WPL * (CH + 1)) % NW, 0);       // (that window was written two windows ago)
            if (it % W == W - 2) {                                     // ... and store it: its buffer was last read in window gw - 1
                const int b2 = (gw + 2) % 3;
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < LPT; ++i) ring[b2 * BUF + q * W * 64 + i * NWV * 64 + tid - lane] = ld[q * LPT + i];
            }
            if (s < 7) {
                const int sn = s + 1;
                sp_split_pair<RELU>(n.h[t], n.m[t], n.l[t], in[sn >> 1][8 * (sn & 1) + 2 * t], in[sn >> 1][8 * (sn & 1) + 2 * t + 1]);
            }
            const sp_bf16x8 bh = sp_op(p.h), bm = sp_op(p.m), bl = sp_op(p.l);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_wop(a3), bh, acc[t], 0, 0, 0);      // small terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_wop(a2), bm, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_wop(a1), bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_wop(a2), bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_wop(a1), bm, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_wop(a1), bh, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (it % W == W - 1) ring_barrier();                       // window closed: every wave has read it, window gw + 2 is in LDS
        }
        p = n;
    }
    if constexpr (RF) {                                                // un-rotate: k-step u's pieces sit in the registers of k-step u - 2
        f32x16 r[4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[u >> 1][8 * (u & 1) + j] = u < 2 ? side[2 * u + (j >> 2)][j & 3] : in[(u - 2) >> 1][8 * ((u - 2) & 1) + j];
#pragma unroll
        for (int t = 0; t < 4; ++t) in[t] = r[t];
    }
}

__global__ __launch_bounds__(512, 2) void k_node_ring(const NodeArgs a) {
    constexpr int NT = 4, L = 128, W = 8, NCH = 6;
    constexpr int BUF = Rn<W>::BUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* ringbase = reinterpret_cast<u32x4*>(smem);
    float* tb = reinterpret_cast<float*>(ringbase + 3 * BUF);
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    // chunks in the order of use: W1 (node part), W1 (aggregate part), W2, W3, WP, WQ
    const u32x4* const src[NCH] = {reinterpret_cast<const u32x4*>(a.split[2]), reinterpret_cast<const u32x4*>(a.split[3]),
                                   reinterpret_cast<const u32x4*>(a.split[0]), reinterpret_cast<const u32x4*>(a.split[1]),
                                   reinterpret_cast<const u32x4*>(a.split[4]), reinterpret_cast<const u32x4*>(a.split[5])};
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int w = 0; w < 2; ++w)                                       // windows 0 and 1 of the first chunk
#pragma unroll
        for (int q = 0; q < 3; ++q) ringbase[w * BUF + q * W * 64 + tid] = src[0][q * 2048 + w * W * 64 + tid];
    __syncthreads();
    // lock-step: every wave of the block runs as many tiles as its wave 0 (the longest walk); padding tiles compute, store nothing
    TileWalk tw0(a.ntiles, 0, MGN_SPREAD_ROUNDS_NODE), tw(a.ntiles, wave, MGN_SPREAD_ROUNDS_NODE);
    if (tw0.tile >= tw0.end) return;
    const int iters = (tw0.end - tw0.tile + tw0.stride - 1) / tw0.stride;
    const int last = tw0.tile + (iters - 1) * tw0.stride;
    auto clamp = [&](int t) { return t < tw.end ? t : last; };
    f32x16 x[NT], acc[NT];
    load_frag<NT>(x, tile_ptr(a.V, clamp(tw.tile), L, lane0), STRIDE_TILE);
    RnFrag nx = rn_read<W>(ringbase + lane0, 0, 0);
    // CSR bounds of this lane's node, one tile ahead (the aggregate's address depends on them)
    auto bounds = [&](int t, int cc, int& b0, int& b1) {
        const int nq = t * TILE + cc;
        const int nc = nq < a.n ? nq : a.n - 1;
        b0 = a.rowptr[nc];
        b1 = nq < a.n ? a.rowptr[nc + 1] : b0;
    };
    int a0, a1;
    bounds(clamp(tw.tile), lane0 & 31, a0, a1);
    for (int j = 0; j < iters; ++j) {
        OPAQUE_LANE();
        const bool on = tw.tile < tw.end;
        const int tile = clamp(tw.tile);
        const int next = clamp(tw.tile + tw.stride);
        const int n = tile * TILE + c;
        const bool valid = on && n < a.n;
        const int nn = n < a.n ? n : 0;
        f32x4* vtile = tile_ptr(a.V, tile, L, lane);
        u32x4* ring = ringbase + lane;
        __builtin_amdgcn_s_setprio(0);
        int b0, b1;
        bounds(next, c, b0, b1);
        // where the aggregate of this node lies (LOAD_AGGREGATE's address logic); its rows arrive during the first chunk
        const int T1 = a0 >> 5, T2 = (a1 - 1) >> 5;
        const int extra = (a1 > a0 && T2 > T1) ? (T2 - T1) : 0;
        const bool from_agg = (a1 > a0) && !extra;
        const f32x4* agg0 = from_agg ? tile_ptr(a.AGG, tile, L, lane) : prow_ptr(a.CARRY, extra ? (int64_t)(2 * T1 + 1) : a.zero_row, L, h);
        const int aggs = from_agg ? STRIDE_TILE : STRIDE_PROW;
        f32x4 side[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) side[m] = agg0[m * aggs];
        tab_frag<NT>(acc, tb + T_B1 * L, h);
#if defined(MGN_WHATIF_NRING) && (MGN_WHATIF_NRING & 4)
        spn_layer<W, 0, NCH, false>(acc, x, ring, src, nx, lane, tid);                     // layer 1, node part
#else
        spn_layer<W, 0, NCH, false, 8, true>(acc, x, ring, src, nx, lane, tid, agg0, aggs, side);   // layer 1, node part; x <- aggregate rows
        for (int q = 1; __any(q <= extra); ++q)                                            // (a receiver whose run straddles more than two edge tiles)
            if (q <= extra) add_frag<NT>(x, prow_ptr(a.CARRY, (int64_t)2 * (T1 + q), L, h), STRIDE_PROW);
#endif
#if defined(MGN_WHATIF_NRING) && (MGN_WHATIF_NRING & 4)     // diagnostic builds (wrong results): what the memory phases cost
        load_frag<NT>(x, tile_ptr(a.V, wave, L, lane), STRIDE_TILE);
#endif
        spn_layer<W, 1, NCH, false>(acc, x, ring, src, nx, lane, tid);                     // layer 1, aggregate part
        tab_frag<NT>(x, tb + T_B2 * L, h);
        spn_layer<W, 2, NCH, true>(x, acc, ring, src, nx, lane, tid);                      // layer 2 (ReLU folded into the split)
        tab_frag<NT>(acc, tb + T_B3 * L, h);
        spn_layer<W, 3, NCH, true>(acc, x, ring, src, nx, lane, tid);                      // layer 3
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_PRIO);
#if defined(MGN_WHATIF_NRING) && (MGN_WHATIF_NRING & 1)
        load_frag<NT>(x, tile_ptr(a.V, wave, L, lane), STRIDE_TILE);
#else
        load_frag<NT>(x, vtile, STRIDE_TILE);                        // v again, for the residual
#endif
        layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);
#pragma unroll
        for (int t = 0; t < NT; ++t) x[t] += acc[t];                 // v <- v + v'
#if defined(MGN_WHATIF_NRING) && (MGN_WHATIF_NRING & 2)
        if (valid && a.n < 0) store_frag<NT>(vtile, STRIDE_TILE, x);
#else
        if (valid) store_frag<NT>(vtile, STRIDE_TILE, x);
#endif
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(0);
        zero_frag<NT>(acc);
        spn_layer<W, 4, NCH, false>(acc, x, ring, src, nx, lane, tid);                     // P = v W1s of the next step's edge MLP
        __builtin_amdgcn_s_setprio(MGN_PRIO);
#if defined(MGN_WHATIF_NRING) && (MGN_WHATIF_NRING & 2)
        if (valid && a.n < 0) store_frag<NT>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, acc);
#else
        if (valid) store_frag<NT>(prow_ptr(a.P, nn, L, h), STRIDE_PROW, acc);
#endif
        __builtin_amdgcn_s_setprio(0);
        tab_frag<NT>(acc, tb + T_BQ * L, h);
#if MGN_NRING_VNEXT
        {   // the next tile's V arrives in the registers this last chunk's input releases: nothing is requested behind the Q stores
            const f32x4* vn = tile_ptr(a.V, next, L, lane);
            f32x4 sv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) sv[m] = vn[m * STRIDE_TILE];
            spn_layer<W, 5, NCH, false, 8, true>(acc, x, ring, src, nx, lane, tid, vn, STRIDE_TILE, sv);   // Q = v W1r + b1
        }
#else
        spn_layer<W, 5, NCH, false>(acc, x, ring, src, nx, lane, tid);                     // Q = v W1r + b1
#endif
        __builtin_amdgcn_s_setprio(MGN_PRIO);
#if defined(MGN_WHATIF_NRING) && (MGN_WHATIF_NRING & 2)
        if (valid && a.n < 0) store_frag<NT>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, acc);
#else
        if (valid) store_frag<NT>(prow_ptr(a.Q, nn, L, h), STRIDE_PROW, acc);
#endif
        PHASE_FENCE();
#if MGN_NRING_VNEXT
#elif defined(MGN_WHATIF_NRING) && (MGN_WHATIF_NRING & 1)
        load_frag<NT>(x, tile_ptr(a.V, wave, L, lane), STRIDE_TILE);
#else
        load_frag<NT>(x, tile_ptr(a.V, next, L, lane), STRIDE_TILE);
#endif
        a0 = b0;
        a1 = b1;
        tw.tile += tw.stride;
    }
}

// ================================================================================================
// Edge step, TWO independent blocks per CU (MGN_FP32_SPLIT=3).  k_edge_ring's eight waves are in lock-step, so the two waves of a SIMD
// are in their epilogues at the same time and the matrix pipe idles for half of the tile; the register-ring kernel (k_edge_split2) has
// waves that overlap freely and drowns in 160 KiB of L2 weight traffic per tile and wave.  Here a block is four waves (one per SIMD)
// with its OWN ring through which ALL pieces pass (72 KiB of LDS: two such blocks fit a CU; 72 KiB of L2 traffic per tile and wave)
// and its own barrier, so the two blocks of a CU drift freely against each other: one's epilogue runs beside the other's chains.  The
// second half of the grid starts half a period late.  No refill machinery: a block's exposed loads are the other block's matrix time.
// ================================================================================================
#ifndef MGN_RING2_PHASE_UNITS
#define MGN_RING2_PHASE_UNITS 7
#endif
__global__ __launch_bounds__(256, 2) void k_edge_ring2(const EdgeArgs a) {
    constexpr int NT = 4, L = 128, W = 8, NCH = 3, NWV = 4;
    constexpr int BUF = Rn<W>::BUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* ringbase = reinterpret_cast<u32x4*>(smem);
    float* tb = reinterpret_cast<float*>(ringbase + 3 * BUF);
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    const u32x4* const src[NCH] = {reinterpret_cast<const u32x4*>(a.split[2]), reinterpret_cast<const u32x4*>(a.split[0]),
                                   reinterpret_cast<const u32x4*>(a.split[1])};      // W1e, W2, W3: three pieces each, 2048 fragments apart
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 8 / NWV; ++i)
                ringbase[w * BUF + q * W * 64 + i * NWV * 64 + tid] = src[0][q * 2048 + w * W * 64 + i * NWV * 64 + tid];
    __syncthreads();
    TileWalk tw0(a.ntiles, 0), tw(a.ntiles, wave);
    if (tw0.tile >= tw0.end) return;
    const int iters = (tw0.end - tw0.tile + tw0.stride - 1) / tw0.stride;
    if (iters >= 16 && blockIdx.x >= gridDim.x / 2)
        for (int i = 0; i < MGN_RING2_PHASE_UNITS; ++i) __builtin_amdgcn_s_sleep(64);
    const int last = a.tile0 + tw0.tile + (iters - 1) * tw0.stride;
    tw.tile += a.tile0;
    tw.end += a.tile0;
    auto clamp = [&](int t) { return t < tw.end ? t : last; };
    f32x16 acc[NT], y[NT];
    EdgeIdx ix = load_edge_idx_nb(a.snd, a.rcv, a.E, clamp(tw.tile), lane0 & 31);
    load_frag<NT>(acc, prow_ptr(a.Q, ix.r >= 0 ? ix.r : 0, L, lane0 >> 5), STRIDE_PROW);
    load_frag<NT>(y, tile_ptr(a.Elat, clamp(tw.tile), L, lane0), STRIDE_TILE);
    RnFrag nx = rn_read<W>(ringbase + lane0, 0, 0);
    for (int j = 0; j < iters; ++j) {
        OPAQUE_LANE();
        const bool on = tw.tile < tw.end;
        const int tile = clamp(tw.tile);
        const int nxt = clamp(tw.tile + tw.stride);
        const EdgeIdx ixn = load_edge_idx_nb(a.snd, a.rcv, a.E, nxt, c);
        const bool valid = on && ix.r >= 0;
        const int r = ix.r >= 0 ? ix.r : 0;
        f32x4* etile = tile_ptr(a.Elat, tile, L, lane);
        u32x4* ring = ringbase + lane;
        __builtin_amdgcn_s_setprio(0);
        spn_layer<W, 0, NCH, false, NWV>(acc, y, ring, src, nx, lane, tid);          // layer 1, edge part (acc entered with Q[r], which carries b1)
        load_frag<NT>(y, prow_ptr(a.P, ix.s, L, h), STRIDE_PROW);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] += y[t];
        tab_frag<NT>(y, tb + T_B2 * L, h);
        spn_layer<W, 1, NCH, true, NWV>(y, acc, ring, src, nx, lane, tid);           // layer 2
        tab_frag<NT>(acc, tb + T_B3 * L, h);
        spn_layer<W, 2, NCH, true, NWV>(acc, y, ring, src, nx, lane, tid);           // layer 3
        PHASE_FENCE();
        __builtin_amdgcn_s_setprio(MGN_PRIO);
        load_frag<NT>(y, etile, STRIDE_TILE);                                         // e again, for the residual
        layer_norm_frag<NT>(acc, tb + T_GAMMA * L, tb + T_BETA * L, h);              // acc = e'
#pragma unroll
        for (int t = 0; t < NT; ++t) y[t] += acc[t];                                  // e <- e + e'
        if (valid) store_frag<NT>(etile, STRIDE_TILE, y);
        load_frag<NT>(y, tile_ptr(a.Elat, nxt, L, lane), STRIDE_TILE);               // the next tile's e
        const int reff = ix.r >= 0 ? r : (-4 - c);
        const int rprev = __shfl_up(reff, 1, 32);
        const int rnext = __shfl_down(reff, 1, 32);
        const bool head = (c == 0) || (reff != rprev);
        const unsigned hm = (unsigned)__ballot(head);
        const int start = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - c))));
        const int st_in = max(start, c & 16);
        const bool c1 = (c - 1 >= st_in), c2 = (c - 2 >= st_in), c4 = (c - 4 >= st_in), c8 = (c - 8 >= st_in);
        const bool cx = (c >= 16) && (start <= 15);
        PHASE_FENCE();
        asm volatile("s_nop 1");
        RG_SCAN_LEVEL(acc, c1, "row_shr:1 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c2, "row_shr:2 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c4, "row_shr:4 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, c8, "row_shr:8 row_mask:0xf bank_mask:0xf");
        RG_SCAN_LEVEL(acc, cx, "row_bcast:15 row_mask:0xa bank_mask:0xf");
        PHASE_FENCE();
        const bool tail = valid && ((c == 31) || (reff != rnext));
        const int r_first = __builtin_amdgcn_readfirstlane(reff);
        const bool sl = (start == 0) && (ix.r_before == r_first);
        const bool sr = (c == 31) && (ix.r_after == reff);
        const bool to_carry = sl || sr;
        f32x4* dst = to_carry ? prow_ptr(a.CARRY, (int64_t)2 * tile + (sl ? 0 : 1), L, h) : tile_ptr(a.AGG, r >> 5, L, 32 * h + (r & 31));
        if (tail) store_frag<NT>(dst, to_carry ? STRIDE_PROW : STRIDE_TILE, acc);
        PHASE_FENCE();
        load_frag<NT>(acc, prow_ptr(a.Q, ixn.r >= 0 ? ixn.r : 0, L, h), STRIDE_PROW);
        ix = ixn;
        tw.tile += tw.stride;
    }
}

