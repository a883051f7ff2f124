// How much VALU / LDS work hides under bf16 MFMAs on gfx950?  (design input for the fp32-accurate "split" kernels:
// six bf16 products per fp32 product, plus ~3.7 VALU instructions per MFMA for the three-way operand split, LayerNorm, scan.)
//   A. ONE wave's own stream: NV instructions of a kind issued behind every MFMA (dependent accumulator chains of 6, as in sp_layer)
//   B. the same at two waves per SIMD
//   C. two waves per SIMD, roles split: waves 0-3 MFMA only, waves 4-7 the same number of VALU instructions alone
// Prints shader cycles (s_memtime) and wall ns (s_memrealtime) per MFMA: the quotient is the clock the part holds.
// Operands are random bit patterns from memory (the clock under load depends on the data).
//   hipcc -O3 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/_overlap_probe && tools/_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { K_FMA = 0, K_SPLIT = 1, K_CVT = 2, K_DPP = 3, K_MAX = 4, K_PERM = 5, K_LDS = 6, K_PKFMA = 7, K_NOP = 8 };
static const char* kind_name[] = {"v_fma_f32", "and+sub (split)", "v_cvt_pk_bf16", "v_fmac_dpp", "v_max_f32", "v_perm_b32", "ds_read_b128", "v_pk_fma_f32", "s_nop 0"};

template <int KIND>
__device__ __forceinline__ void filler(float (&r)[16], f32x2 (&p)[4], f32x4& l, int i, const f32x4* lds, float m) {
    float& a = r[i & 15];
    float& b = r[(i + 5) & 15];
    if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(b));
    if constexpr (KIND == K_SPLIT) {
        if (i & 1) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a) : "v"(b));
        else asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a) : "v"(b));
    }
    if constexpr (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(m));
    if constexpr (KIND == K_DPP) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a) : "v"(m));
    if constexpr (KIND == K_MAX) asm volatile("v_max_f32 %0, 0, %1" : "=v"(a) : "v"(b));
    if constexpr (KIND == K_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(m), "v"(r[(i + 9) & 15]));
    if constexpr (KIND == K_LDS) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"((int)(size_t)lds + (i & 63) * 1024));
    if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]));
    if constexpr (KIND == K_NOP) asm volatile("s_nop 0");
}

// SHAPE 0: v_mfma_f32_32x32x16_bf16 (16 accumulator registers), 1: v_mfma_f32_16x16x32_bf16 (4)
template <int SHAPE, int KIND, int NV, bool MF>
__device__ __forceinline__ void body(int iters, const f32x4* src, float* sink, const f32x4* lds, unsigned long long* tc, unsigned long long* tr) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    f32x4 acc4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
        acc4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 a[3], b[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a[k] = src[(k * 64 + lane) & 1023];
        b[k] = src[((k + 3) * 64 + lane) & 1023];
    }
    float r[16];
    f32x2 p[4];
    f32x4 l = a[0];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = a[k % 3][k & 3] * 1e-3f + k;
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = f32x2{r[k], r[k + 4]};
    const float m = 1.0001f;
    unsigned long long c0, c1, w0, w1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(w0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 6; ++q) {          // six dependent products per accumulator block, as in sp_layer
                if constexpr (MF) {
                    if constexpr (SHAPE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a[q % 3]), "v"(b[q / 2]));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[t]) : "v"(a[q % 3]), "v"(b[q / 2]));
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) filler<KIND>(r, p, l, (t * 6 + q) * NV + v, lds, m);
            }
        if constexpr (KIND == K_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(w1)::"memory");
    float s = l[0] + l[3];
#pragma unroll
    for (int t = 0; t < 4; ++t) s += acc[t][t] + acc4[t][t & 3];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += r[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) s += p[k][0] + p[k][1];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) {
        tc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0;
        tr[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w1 - w0;
    }
}

// ROLE 0: every wave runs MFMA + fillers; 1: waves 0-3 MFMA only, waves 4-7 fillers only (8-wave blocks)
template <int SHAPE, int KIND, int NV, int ROLE>
__global__ __launch_bounds__(512, 1) void k_probe(int iters, const f32x4* src, float* sink, unsigned long long* tc, unsigned long long* tr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = src[i & 1023];
    __syncthreads();
    const f32x4* mine = lds + (threadIdx.x & 63);
    if constexpr (ROLE == 0) body<SHAPE, KIND, NV, true>(iters, src, sink, mine, tc, tr);
    else if ((threadIdx.x >> 6) < 4) body<SHAPE, K_NOP, 0, true>(iters, src, sink, mine, tc, tr);
    else body<SHAPE, KIND, NV, false>(iters, src, sink, mine, tc, tr);
}

static f32x4* d_src;
static float* d_sink;
static unsigned long long *d_tc, *d_tr;

static double med(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

template <int SHAPE, int KIND, int NV, int ROLE>
void run(int waves, int iters) {
    const int nb = 256, nw = nb * waves;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_probe<SHAPE, KIND, NV, ROLE>), dim3(nb), dim3(64 * waves), 65536, 0, iters, d_src, d_sink, d_tc, d_tr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> tc(nw), tr(nw);
    hipMemcpy(tc.data(), d_tc, nw * 8, hipMemcpyDeviceToHost);
    hipMemcpy(tr.data(), d_tr, nw * 8, hipMemcpyDeviceToHost);
    const double nm = 24.0 * iters;
    if (ROLE == 0) {
        std::vector<double> c, w;
        for (int i = 0; i < nw; ++i) { c.push_back(tc[i] / nm); w.push_back(tr[i] * 10.0 / nm); }
        const double cyc = med(c), ns = med(w);
        const double flop = SHAPE == 0 ? 32768.0 : 16384.0;
        printf("%s %-16s NV=%d  %d wave/SIMD: %6.1f cyc, %6.2f ns per MFMA per wave (clock %.2f GHz)  -> %6.0f TFLOP/s bf16\n", SHAPE ? "16x16x32" : "32x32x16",
               kind_name[KIND], NV, waves / 4, cyc, ns, cyc / ns, flop * (waves / 4) * 1024 / ns * 1e-3);
    } else {
        std::vector<double> cm, cv, wm, wv;
        for (int i = 0; i < nw; ++i) {
            if ((i % waves) < 4) { cm.push_back(tc[i] / nm); wm.push_back(tr[i] * 10.0 / nm); }
            else { cv.push_back(tc[i] / nm); wv.push_back(tr[i] * 10.0 / nm); }
        }
        printf("%s %-16s NV=%d  split roles: MFMA wave %6.1f cyc / %6.2f ns per MFMA; VALU wave %6.1f cyc / %6.2f ns per %d instr\n", SHAPE ? "16x16x32" : "32x32x16",
               kind_name[KIND], NV, med(cm), med(wm), med(cv), med(wv), NV);
    }
}

#define SWEEP(SHAPE, KIND)                \
    run<SHAPE, KIND, 2, 0>(4, iters);     \
    run<SHAPE, KIND, 4, 0>(4, iters);     \
    run<SHAPE, KIND, 6, 0>(4, iters);     \
    run<SHAPE, KIND, 8, 0>(4, iters);     \
    run<SHAPE, KIND, 4, 0>(8, iters);     \
    run<SHAPE, KIND, 6, 0>(8, iters);     \
    run<SHAPE, KIND, 4, 1>(8, iters);     \
    run<SHAPE, KIND, 6, 1>(8, iters);

int main() {
    const int iters = 2000;
    std::vector<uint32_t> h(4096);
    uint64_t st = 88172645463325252ull;
    for (auto& x : h) {          // random bf16 pairs of moderate magnitude (exponent 120..127)
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const uint32_t lo = (uint32_t)st, e1 = 120 + (lo & 7), e2 = 120 + ((lo >> 3) & 7);
        x = ((lo & 0x80000000u) | (e1 << 23) | ((lo >> 8) & 0x7F0000u)) | ((((lo >> 6) & 0x8000u) | (e2 << 7) | ((lo >> 20) & 0x7Fu)) & 0xFFFFu);
    }
    hipMalloc(&d_src, 4096 * 4);
    hipMemcpy(d_src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_sink, 256 * 512 * 4);
    hipMalloc(&d_tc, 256 * 8 * 8);
    hipMalloc(&d_tr, 256 * 8 * 8);
    // warm the clocks
    for (int i = 0; i < 20; ++i) run<0, K_NOP, 0, 0>(8, iters), (void)0;
    printf("---- MFMA alone\n");
    run<0, K_NOP, 0, 0>(4, iters);
    run<0, K_NOP, 0, 0>(8, iters);
    run<1, K_NOP, 0, 0>(4, iters);
    run<1, K_NOP, 0, 0>(8, iters);
    run<0, K_NOP, 1, 0>(4, iters);
    run<0, K_NOP, 3, 0>(4, iters);
    printf("---- 32x32x16 with fillers\n");
    SWEEP(0, K_FMA)
    SWEEP(0, K_SPLIT)
    SWEEP(0, K_CVT)
    SWEEP(0, K_DPP)
    SWEEP(0, K_MAX)
    SWEEP(0, K_PERM)
    SWEEP(0, K_PKFMA)
    run<0, K_LDS, 1, 0>(4, iters);
    run<0, K_LDS, 2, 0>(4, iters);
    run<0, K_LDS, 1, 0>(8, iters);
    run<0, K_LDS, 2, 0>(8, iters);
    printf("---- 16x16x32 with fillers (half the flops per MFMA)\n");
    run<1, K_FMA, 1, 0>(4, iters);
    run<1, K_FMA, 2, 0>(4, iters);
    run<1, K_FMA, 3, 0>(4, iters);
    run<1, K_FMA, 2, 0>(8, iters);
    run<1, K_FMA, 3, 0>(8, iters);
    run<1, K_SPLIT, 2, 0>(4, iters);
    run<1, K_SPLIT, 3, 0>(4, iters);
    run<1, K_SPLIT, 2, 0>(8, iters);
    run<1, K_LDS, 1, 0>(4, iters);
    run<1, K_LDS, 1, 0>(8, iters);
    return 0;
}
