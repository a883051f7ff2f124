// NOT BUILT.  k_edge_ring on v_mfma_f32_16x16x32_bf16 (three bf16 pieces), opt-in in round 4 (MGN_EDGE_RING16=1), removed from csrc/split.hip
// in round 5: parity green; edge kernel + 2.8 %, the node kernels behind it - 6 % (clock), step equal (docs/experiments.md
// round 4).  Uses the helpers of csrc/split.hip / split_common.hpp (N16Buf, Rg, ring_barrier, RG_SCAN_LEVEL ...) and the weight pieces
// in the 16x16x32 fragment order (EdgeArgs.split16, still packed for the 16-row kernels).
// ================================================================================================
// k_edge_ring on v_mfma_f32_16x16x32_bf16 (MGN_FP32_SPLIT=4).  The same lock-step ring, the same turnover (next tile's e through layer
// 3's refill), the same memory layouts -- another register layout, because the part holds a higher clock on this MFMA shape under the
// same matrix load (docs/experiments.md round 4: the what-if build with the shape swapped in place runs the edge kernel 2.4 % and the
// node kernels behind it 4 % faster).  "N16" fragment: lane (r16 = l & 15, g = l >> 4) owns, for the two row blocks rb (rows 16 rb + r16)
// and the eight feature blocks fb, the four features 16 fb + 4 g + i -- the float4 number 4 fb + g of the row, so tables, P / Q rows and
// tile-major pieces are addressed with the row layouts' own piece arithmetic.  A 16 x 16 accumulator block (ob, rb) has that very
// layout, and k-step ks (32 inputs) of the next layer takes blocks 2 ks and 2 ks + 1 as its B operand: element j of lane group g is input
// 16 (2 ks + (j >> 2)) + 4 g + (j & 3), which is how the host orders the A fragments (mgn_api.cpp: pack_chunk16_bf16).  A step
// (ks, ob) = three fragments (hi resident, mid / lo from the ring), twelve MFMAs (six products x two row blocks): 32 steps per layer
// as before, so the ring's windows, requests and barriers are k_edge_ring's.
// ================================================================================================
#ifndef MGN_R16_RFKS
#define MGN_R16_RFKS 2      // k-steps of layer 3 whose released blocks take the next tile's e inside the layer (the rest at the start of the epilogue)
#endif
#ifndef MGN_R16_ILV
#define MGN_R16_ILV 0
#endif
#ifndef MGN_R16_WHATIF
#define MGN_R16_WHATIF 0    // timing-only builds: 1 no e stores, 2 no tail stores, 4 no scan, 8 no re-read of e (wrong results)
#endif
#ifndef MGN_R16_EPI_FENCE
#define MGN_R16_EPI_FENCE 0
#endif
#ifndef MGN_R16_LN_FENCE
#define MGN_R16_LN_FENCE 0
#endif
#ifndef MGN_R16_TAIL
#define MGN_R16_TAIL 1      // 1: the rest at the END of the epilogue (behind its stores; 0: at its start -- 32 registers more through the epilogue, spills)
#endif
struct Sp16Pieces { unsigned h[2][4], m[2][4], l[2][4]; };   // [row block][dword]: the pieces of one k-step (8 bf16 per lane and row block)

// piece fb of a tile-major fragment array / of a P-layout row: p[fb * N16_TILE_FB] / p[fb * N16_PROW_FB]
constexpr int N16_TILE_FB = 2 * STRIDE_TILE, N16_PROW_FB = 2 * STRIDE_PROW;
DEVINL f32x4* n16_tile_ptr(float* base, int64_t tile, int L, int r16, int g, int rb) {
    return reinterpret_cast<f32x4*>(base + tile * (TILE * L)) + (g >> 1) * STRIDE_TILE + 32 * (g & 1) + 16 * rb + r16;
}
DEVINL const f32x4* n16_tile_ptr(const float* base, int64_t tile, int L, int r16, int g, int rb) {
    return reinterpret_cast<const f32x4*>(base + tile * (TILE * L)) + (g >> 1) * STRIDE_TILE + 32 * (g & 1) + 16 * rb + r16;
}
DEVINL f32x4* n16_prow_ptr(float* base, int64_t row, int L, int g) { return reinterpret_cast<f32x4*>(base) + prow_index(row, L) + (g >> 1) * STRIDE_PROW + (g & 1); }
DEVINL const f32x4* n16_prow_ptr(const float* base, int64_t row, int L, int g) {
    return reinterpret_cast<const f32x4*>(base) + prow_index(row, L) + (g >> 1) * STRIDE_PROW + (g & 1);
}

// every global access of the kernel goes through a buffer descriptor: a wave-uniform 64-bit base (scalar registers) + a 32-bit byte
// offset per lane + a scalar / immediate block offset -- ONE address register per stream where 64-bit pointers cost four pairs (a
// block is 2 KiB, the immediate of global_load reaches 4): the epilogue holds 160 data registers and has none to spare for addresses.
// (launch_edge_step runs the kernel only where every array is shorter than 4 GiB: EdgeArgs::off32.)
#define N16_LD(BUF, SOFF, OFF) n16_ld(BUF, OFF, SOFF)
#define N16_ST(BUF, SOFF, OFF, V) n16_st(BUF, OFF, SOFF, V)
// One L x L layer.  RF = 1: `in` is refilled with block fb of the stream (*rfb at fb * RFS * 16, lane offsets rfo0 / rfo1 per row block) rotated by one k-step (the pieces of blocks 0, 1 wait in side[rb][0 .. 1],
// requested before the layer; `in` comes back holding the rows); RF = 2: in place, the blocks of k-steps 0 .. RFKS - 1 inside the layer, the
// rest left to the caller (requested inside the layer they are spilled where they land: the 64 registers are free only at its end).
template <int W, int LYR, bool RELU, int RF, int RFS, int NWV, int RFKS = 3>
DEVINL void sp16_layer_ring(f32x4 (&acc)[2][8], f32x4 (&in)[2][8], const u32x4* hi, const u32x4* hi_next, u32x4* ring, const RingSrc& src,
                            RingFrag& nx, int lane, int tid, const N16Buf* rfb = nullptr, unsigned rfo0 = 0, unsigned rfo1 = 0, f32x4 (*side)[2] = nullptr) {
    constexpr int WPL = Rg<W>::WPL, NW = Rg<W>::NW, BUF = Rg<W>::BUF;
    constexpr int LPT = 8 / NWV;
    Sp16Pieces p;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int u = 0; u < 4; ++u) sp_split_pair<RELU>(p.h[rb][u], p.m[rb][u], p.l[rb][u], in[rb][u >> 1][2 * (u & 1)], in[rb][u >> 1][2 * (u & 1) + 1]);
    u32x4 ld_m[LPT], ld_l[LPT];
    unsigned voff = (unsigned)tid * 16u;
    asm volatile("" : "+v"(voff));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        Sp16Pieces n;
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) {
            const int it = 8 * ks + ob;
            const int gw = WPL * LYR + it / W;
            const u32x4 a1 = nx.h, a2 = nx.m, a3 = nx.l;
            if (it % W == 0) {                                         // request window gw + 2
                const int g2 = (gw + 2) % NW, l2 = g2 / WPL, w2 = g2 % WPL;
#pragma unroll
                for (int i = 0; i < LPT; ++i) {
                    ld_m[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src.mid[l2] + w2 * W * 64 + i * NWV * 64) + voff);
                    ld_l[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src.lo[l2] + w2 * W * 64 + i * NWV * 64) + voff);
                }
            }
            if constexpr (RF != 0) {                                   // one request every second step: the blocks k-step ks has released
                if ((ob & 1) && ks < (RF == 1 ? 3 : RFKS)) {
                    const int rb = (ob >> 1) & 1, q = ob >> 2;         // which of the two blocks of this k-step
                    const int fb_src = RF == 1 ? 2 * (ks + 1) + q : 2 * ks + q;
                    in[rb][2 * ks + q] = N16_LD(*rfb, fb_src * RFS * 16, rb ? rfo1 : rfo0);
                }
            }
            if (it + 1 < 32) {
                const int gn = WPL * LYR + (it + 1) / W;
                nx.h = hi[(it + 1) * 64 + lane];
                nx.m = ring[(gn % 3) * BUF + ((it + 1) % W) * 64];
                nx.l = ring[(gn % 3) * BUF + W * 64 + ((it + 1) % W) * 64];
            } else if (LYR < 2) {
                nx = ring_first<W, (LYR + 1) % 3>(hi_next, ring, lane);
            }
            if (it % W == W - 2) {
                const int b2 = (gw + 2) % 3;
#pragma unroll
                for (int i = 0; i < LPT; ++i) {
                    ring[b2 * BUF + i * NWV * 64 + tid - lane] = ld_m[i];
                    ring[b2 * BUF + W * 64 + i * NWV * 64 + tid - lane] = ld_l[i];
                }
            }
            if (ks < 3) {                                              // one pair of the next k-step's pieces per step
                const int kn = ks + 1, rb = ob >> 2, u = ob & 3;
                sp_split_pair<RELU>(n.h[rb][u], n.m[rb][u], n.l[rb][u], in[rb][2 * kn + (u >> 1)][2 * (u & 1)], in[rb][2 * kn + (u >> 1)][2 * (u & 1) + 1]);
            }
#if MGN_R16_ILV
            {   // the two row blocks' chains interleaved: no MFMA waits for the one issued just before it
                const sp_bf16x8 bh0 = sp_op(p.h[0]), bm0 = sp_op(p.m[0]), bl0 = sp_op(p.l[0]);
                const sp_bf16x8 bh1 = sp_op(p.h[1]), bm1 = sp_op(p.m[1]), bl1 = sp_op(p.l[1]);
                acc[0][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a3), bh0, acc[0][ob], 0, 0, 0);   // small terms first
                acc[1][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a3), bh1, acc[1][ob], 0, 0, 0);
                acc[0][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a2), bm0, acc[0][ob], 0, 0, 0);
                acc[1][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a2), bm1, acc[1][ob], 0, 0, 0);
                acc[0][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a1), bl0, acc[0][ob], 0, 0, 0);
                acc[1][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a1), bl1, acc[1][ob], 0, 0, 0);
                acc[0][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a2), bh0, acc[0][ob], 0, 0, 0);
                acc[1][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a2), bh1, acc[1][ob], 0, 0, 0);
                acc[0][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a1), bm0, acc[0][ob], 0, 0, 0);
                acc[1][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a1), bm1, acc[1][ob], 0, 0, 0);
                acc[0][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a1), bh0, acc[0][ob], 0, 0, 0);
                acc[1][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a1), bh1, acc[1][ob], 0, 0, 0);
            }
#else
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const sp_bf16x8 bh = sp_op(p.h[rb]), bm = sp_op(p.m[rb]), bl = sp_op(p.l[rb]);
                acc[rb][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a3), bh, acc[rb][ob], 0, 0, 0);   // small terms first
                acc[rb][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a2), bm, acc[rb][ob], 0, 0, 0);
                acc[rb][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a1), bl, acc[rb][ob], 0, 0, 0);
                acc[rb][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a2), bh, acc[rb][ob], 0, 0, 0);
                acc[rb][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a1), bm, acc[rb][ob], 0, 0, 0);
                acc[rb][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sp_wop(a1), bh, acc[rb][ob], 0, 0, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (it % W == W - 1) ring_barrier();
        }
        p = n;
    }
    if constexpr (RF == 1) {                                           // un-rotate: block fb's piece sits in the registers of block fb - 2
        f32x4 r[2][8];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) r[rb][fb] = fb < 2 ? side[rb][fb] : in[rb][fb - 2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) in[rb][fb] = r[rb][fb];
    }
}

// sum over the four lane groups (lanes l, l ^ 16, l ^ 32, l ^ 48), the same value in all of them.  (v_permlane16 / 32_swap instead of the
// two LDS-crossbar round trips, and the scan's carry between the row blocks as a DPP row rotation instead of 32 ds_swizzle: built, 3.198
// vs 3.203 ms -- nothing -- and taken out again.)
DEVINL float n16_sum4(float x) {
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}
#define RG16_SCAN_LEVEL(ACC, COND, CTRL)                                                                                     \
    do {                                                                                                                     \
        const float m_ = (COND) ? 1.f : 0.f;                                                                                 \
        _Pragma("unroll") for (int f_ = 0; f_ < 8; ++f_)                                                                     \
            _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_)                                                                 \
                asm volatile("v_fmac_f32_dpp %0, %0, %1 " CTRL " bound_ctrl:0" : "+v"(ACC[f_][k_]) : "v"(m_));               \
    } while (0)

template <int NWV>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void k_edge_ring16(const EdgeArgs a) {
    constexpr int L = 128, PC = 16384;
    constexpr int W = 8;
    constexpr int BUF = Rg<W>::BUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    copy_to_lds16(wl, a.split16[2], PC, true);                       // hi of W1e, W2, W3
    copy_to_lds16(wl + PC, a.split16[0], PC, true);
    copy_to_lds16(wl + 2 * PC, a.split16[1], PC, true);
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
    u32x4* ringbase = reinterpret_cast<u32x4*>(wl + 3 * PC);
    float* tb = reinterpret_cast<float*>(ringbase + 3 * BUF);
    copy_to_lds(tb, a.tabs, T_COUNT * L);
    RingSrc src;
    {
        const u32x4* gsrc[3] = {reinterpret_cast<const u32x4*>(a.split16[2]), reinterpret_cast<const u32x4*>(a.split16[0]),
                                reinterpret_cast<const u32x4*>(a.split16[1])};
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            src.mid[l] = gsrc[l] + 2048;
            src.lo[l] = gsrc[l] + 4096;
        }
#pragma unroll
        for (int w = 0; w < 2; ++w)
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
    TileWalk tw0(a.ntiles, 0), tw(a.ntiles, wave);
    if (tw0.tile >= tw0.end) return;
    const int iters = (tw0.end - tw0.tile + tw0.stride - 1) / tw0.stride;
    if (MGN_RING_PHASES > 1 && iters >= 32) {
        const int ph = (int)(blockIdx.x / NUM_XCD) % MGN_RING_PHASES;
        for (int i = 0; i < ph * MGN_RING_PHASE_UNITS; ++i) __builtin_amdgcn_s_sleep(64);
    }
    const int last = a.tile0 + tw0.tile + (iters - 1) * tw0.stride;
    tw.tile += a.tile0;
    tw.end += a.tile0;
    auto clamp = [&](int t) { return t < tw.end ? t : last; };
    f32x4 acc[2][8], y[2][8], side[2][2];
    // sender / receiver of this lane's two rows (-1 marks a padding row); the receivers around the tile are wave-uniform (scalar)
    auto load_sr = [&](int t, int c, int& s_, int& r_) {
        const int64_t eid = (int64_t)t * TILE + c;
        const bool v = eid < a.E;
        const int64_t ec = v ? eid : a.E - 1;
        const int sv = a.snd[ec], rv = a.rcv[ec];
        s_ = v ? sv : 0;
        r_ = v ? rv : -1;
    };
    constexpr int TFB = N16_TILE_FB * 16, PFB = N16_PROW_FB * 16;      // bytes from one feature block of a lane to the next
    auto tile_buf = [&](int t) { return n16_buf(a.Elat + (int64_t)t * (TILE * L), TILE * L * 4); };                                         // (uniform)
    auto tile_off = [](int r16, int g) { return (unsigned)(((g >> 1) * STRIDE_TILE + 32 * (g & 1) + r16) * 16); };              // row block 1: + 256
    auto prow_off = [&](int row, int g) { return (unsigned)(prow_index(row, L) * 16) + (unsigned)(((g >> 1) * STRIDE_PROW + (g & 1)) * 16); };
    const N16Buf Pb = n16_buf(a.P), Qb = n16_buf(a.Q), Ab = n16_buf(a.AGG), Cb = n16_buf(a.CARRY);
    int is[2], ir[2];
    {
        const int r16 = lane0 & 15, g0 = lane0 >> 4;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            load_sr(clamp(tw.tile), 16 * rb + r16, is[rb], ir[rb]);
            const unsigned qo = prow_off(ir[rb] >= 0 ? ir[rb] : 0, g0), po = prow_off(is[rb], g0), eo = tile_off(r16, g0) + 256u * rb;
            const N16Buf eb = tile_buf(clamp(tw.tile));
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) {
                acc[rb][fb] = N16_LD(Qb, fb * PFB, qo);
                y[rb][fb] = N16_LD(eb, fb * TFB, eo);
            }
            side[rb][0] = N16_LD(Pb, 0, po);
            side[rb][1] = N16_LD(Pb, PFB, po);
        }
    }
    int stamp_tile = 0;
    (void)stamp_tile;
    for (int j = 0; j < iters; ++j, ++stamp_tile) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int r16 = lane & 15, g = lane >> 4;
        const bool on = tw.tile < tw.end;
        const int tile = clamp(tw.tile);
        const int nxt = clamp(tw.tile + tw.stride);
        u32x4* ring = ringbase + lane;
        __builtin_amdgcn_s_setprio(0);
        STAMP(0);
        RingFrag nx = ring_first<W, 0>(l1h, ring, lane);
        // layer 1 (edge part): y = e tile in, P[s] out (acc entered with Q[r], which carries b1)
        sp16_layer_ring<W, 0, false, 1, N16_PROW_FB, NWV>(acc, y, l1h, l2h, ring, src, nx, lane, tid, &Pb, prow_off(is[0], g), prow_off(is[1], g), side);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) acc[rb][fb] += y[rb][fb];
        CST(1);
        {
            const f32x4* t4 = reinterpret_cast<const f32x4*>(tb + T_B2 * L) + g;
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) y[0][fb] = y[1][fb] = t4[4 * fb];
        }
        CST(2);
        sp16_layer_ring<W, 1, true, 0, 0, NWV>(y, acc, l2h, l3h, ring, src, nx, lane, tid);      // layer 2 (ReLU folded into the split)
        CST(3);
        {
            const f32x4* t4 = reinterpret_cast<const f32x4*>(tb + T_B3 * L) + g;
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) acc[0][fb] = acc[1][fb] = t4[4 * fb];
        }
        CST(4);
        const N16Buf ecb = tile_buf(tile), enb = tile_buf(nxt);
        // layer 3: y = layer 2's output in, the NEXT tile's e out (blocks 0 .. 5; 6, 7 at the start of the epilogue)
        sp16_layer_ring<W, 2, true, 2, N16_TILE_FB, NWV, MGN_R16_RFKS>(acc, y, l3h, l1h, ring, src, nx, lane, tid, &enb, tile_off(r16, g),
                                                                       tile_off(r16, g) + 256u);
        PHASE_FENCE();
        CST(5);
        __builtin_amdgcn_s_setprio(MGN_PRIO);
        // the indices the epilogue needs -- the next tile's rows, the receivers either side of this tile -- are requested here, not at
        // the top of the tile: nothing index-shaped stays live across the three layers (the allocator spilled them where they landed,
        // with a full s_waitcnt vmcnt(0) behind the tile's operand requests)
        int isn[2], irn[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) load_sr(nxt, 16 * rb + r16, isn[rb], irn[rb]);
        const int64_t e0 = (int64_t)tile * TILE;
        const int rbv = a.rcv[e0 > 0 ? e0 - 1 : 0], rav = a.rcv[e0 + TILE < a.E ? e0 + TILE : a.E - 1];
        f32x4 er[2][8];              // this tile's e again, for the residual
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) er[rb][fb] = (MGN_R16_WHATIF & 8) ? y[rb][fb & 3] : N16_LD(ecb, fb * TFB + 256 * rb, tile_off(r16, g));
#if MGN_R16_TAIL != 1
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)       // the blocks of the next tile's e that layer 3 did not take (2: the first half of them)
#pragma unroll
            for (int fb = 2 * MGN_R16_RFKS; fb < (MGN_R16_TAIL == 2 ? MGN_R16_RFKS + 4 : 8); ++fb)
                y[rb][fb] = N16_LD(enb, fb * TFB + 256 * rb, tile_off(r16, g));
#endif
#if MGN_R16_EPI_FENCE
        PHASE_FENCE();               // every request of the epilogue is in the queue before its first store (s_waitcnt vmcnt counts in order)
#endif
        {   // LayerNorm per row: 32 of a row's features in this lane, the rest in the lanes r16 + 16 g'
            constexpr float invL = 1.0f / 128;
            float rstd[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                float sm = 0.f;
#pragma unroll
                for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) sm += acc[rb][fb][i];
                sm = n16_sum4(sm);
                const float mean = sm * invL;
                float q = 0.f;
#pragma unroll
                for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float d = acc[rb][fb][i] - mean;
                        acc[rb][fb][i] = d;
                        q += d * d;
                    }
                q = n16_sum4(q);
                rstd[rb] = ln_rstd_at(q * invL, tb + T_LN * L);
            }
            const f32x4* g4 = reinterpret_cast<const f32x4*>(tb + T_GAMMA * L) + g;
            const f32x4* b4 = reinterpret_cast<const f32x4*>(tb + T_BETA * L) + g;
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) {                                 // one block of gamma / beta at a time, for both row blocks
                const f32x4 gv = g4[4 * fb], bv = b4[4 * fb];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[rb][fb][i] = acc[rb][fb][i] * rstd[rb] * gv[i] + bv[i];
#if MGN_R16_LN_FENCE
                if (fb & 1) PHASE_FENCE();
#endif
            }
        }
        CST(6);
        bool valid[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            valid[rb] = on && ir[rb] >= 0;
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) er[rb][fb] += acc[rb][fb];                        // e <- e + e'
            {   // padding rows (and a wave past the end of its walk) store nowhere: out of the descriptor's range, so that no branch
                // stands between the requests of this tile's e and their use (the compiler sinks them into it)
                const unsigned eo = (valid[rb] && !(MGN_R16_WHATIF & 1)) ? tile_off(r16, g) : N16_DROP;
#pragma unroll
                for (int fb = 0; fb < 8; ++fb) N16_ST(ecb, fb * TFB + 256 * rb, eo, er[rb][fb]);
            }
        }
        CST(7);
        // ---- segmented sum of e' over runs of equal receiver: rows 0 .. 15 in block 0, 16 .. 31 in block 1, one DPP row per lane group
        const int cA = r16, cB = 16 + r16;
        const int reffA = ir[0] >= 0 ? ir[0] : (-4 - cA), reffB = ir[1] >= 0 ? ir[1] : (-4 - cB);
        const int prevA = __shfl_up(reffA, 1, 16);
        const int prevBin = __shfl_up(reffB, 1, 16);
        const int lastA = __shfl(reffA, 15, 16), firstB = __shfl(reffB, 0, 16);
        const int nextA = __shfl_down(reffA, 1, 16), nextB = __shfl_down(reffB, 1, 16);
        const bool headA = (r16 == 0) || (reffA != prevA);
        const bool headB = reffB != (r16 == 0 ? lastA : prevBin);
        const unsigned hm = ((unsigned)__ballot(headA) & 0xFFFFu) | (((unsigned)__ballot(headB) & 0xFFFFu) << 16);
        const int startA = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - cA))));
        const int startB = 31 - __clz((int)(hm & (0xFFFFFFFFu >> (31 - cB))));
        const int stB = max(startB, 16);
        const bool a1 = (cA - 1 >= startA), a2 = (cA - 2 >= startA), a4 = (cA - 4 >= startA), a8 = (cA - 8 >= startA);
        const bool b1 = (cB - 1 >= stB), b2 = (cB - 2 >= stB), b4 = (cB - 4 >= stB), b8 = (cB - 8 >= stB);
        const bool bx = startB <= 15;
        PHASE_FENCE();
        asm volatile("s_nop 1");
        RG16_SCAN_LEVEL(acc[0], a1, "row_shr:1 row_mask:0xf bank_mask:0xf");
        RG16_SCAN_LEVEL(acc[1], b1, "row_shr:1 row_mask:0xf bank_mask:0xf");
        RG16_SCAN_LEVEL(acc[0], a2, "row_shr:2 row_mask:0xf bank_mask:0xf");
        RG16_SCAN_LEVEL(acc[1], b2, "row_shr:2 row_mask:0xf bank_mask:0xf");
        RG16_SCAN_LEVEL(acc[0], a4, "row_shr:4 row_mask:0xf bank_mask:0xf");
        RG16_SCAN_LEVEL(acc[1], b4, "row_shr:4 row_mask:0xf bank_mask:0xf");
        RG16_SCAN_LEVEL(acc[0], a8, "row_shr:8 row_mask:0xf bank_mask:0xf");
        RG16_SCAN_LEVEL(acc[1], b8, "row_shr:8 row_mask:0xf bank_mask:0xf");
        PHASE_FENCE();
        {   // a run that crosses from row 15 into block 1 takes the total of row 15 (lane 15 of the lane group) along
            const float mx = bx ? 1.f : 0.f;
#pragma unroll
            for (int fb = 0; fb < 8; ++fb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t15 = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, (float)acc[0][fb][i]), 0x1F0));
                    acc[1][fb][i] = fmaf(t15, mx, acc[1][fb][i]);
                }
        }
        const int r_first = __builtin_amdgcn_readfirstlane(reffA);
        const int r_before = tile > 0 ? __builtin_amdgcn_readfirstlane(rbv) : -2;
        const int r_after = e0 + TILE < a.E ? __builtin_amdgcn_readfirstlane(rav) : -3;
        const bool tailA = valid[0] && (reffA != (r16 == 15 ? firstB : nextA));
        const bool tailB = valid[1] && ((cB == 31) || (reffB != nextB));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int c = rb ? cB : cA, start = rb ? startB : startA, reff = rb ? reffB : reffA;
            const bool tail = rb ? tailB : tailA;
            const bool sl = (start == 0) && (r_before == r_first);
            const bool sr = (c == 31) && (r_after == reff);
            const bool to_carry = sl || sr;
            const int r = ir[rb] >= 0 ? ir[rb] : 0;
            const unsigned dst = to_carry ? prow_off(2 * tile + (sl ? 0 : 1), g) : (unsigned)(r >> 5) * (unsigned)(TILE * L * 4) + tile_off(r & 31, g);
            if (tail && !(MGN_R16_WHATIF & 2)) {
                if (to_carry) {
#pragma unroll
                    for (int fb = 0; fb < 8; ++fb) N16_ST(Cb, fb * PFB, dst, acc[rb][fb]);
                } else {
#pragma unroll
                    for (int fb = 0; fb < 8; ++fb) N16_ST(Ab, fb * TFB, dst, acc[rb][fb]);
                }
            }
        }
        PHASE_FENCE();
        // turnover: the next tile's layer-1 accumulator starts from Q[r]; the first two blocks of its P rows wait in `side`
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const unsigned qo = prow_off(irn[rb] >= 0 ? irn[rb] : 0, g), po = prow_off(isn[rb], g);
#pragma unroll
            for (int fb = 0; fb < 8; ++fb) acc[rb][fb] = N16_LD(Qb, fb * PFB, qo);
            side[rb][0] = N16_LD(Pb, 0, po);
            side[rb][1] = N16_LD(Pb, PFB, po);
            is[rb] = isn[rb];
            ir[rb] = irn[rb];
#if MGN_R16_TAIL
#pragma unroll                                                          // the blocks of the next tile's e that layer 3 did not take
            for (int fb = (MGN_R16_TAIL == 2 ? MGN_R16_RFKS + 4 : 2 * MGN_R16_RFKS); fb < 8; ++fb) y[rb][fb] = N16_LD(enb, fb * TFB + 256 * rb, tile_off(r16, g));
#endif
        }
        tw.tile += tw.stride;
    }
}

