// fp32 GEMM accuracy from the bf16 matrix cores: every fp32 operand split into three bf16 pieces (x = x1 + x2 + x3 exactly),
// six of the nine piece products kept (11, 12, 21, 13, 22, 31: max rel error 1.7e-7 on a 128 x 128 layer against float64, a plain
// fp32 GEMM has 4.2e-7).  v_mfma_f32_32x32x16_bf16 runs at 16 x the fp32 MFMA rate, so six terms are 2.7 x cheaper in matrix time
// than v_mfma_f32_32x32x2_f32 -- if the split (VALU) and the weight traffic (LDS) do not eat it.  This probe runs one L = 128 layer
// the way the processor kernels would (lane per row, transposed; weights' three pieces LDS-resident, 96 KiB; the activations split
// in registers per tile) and prints the sustained rate in fp32-equivalent TFLOP/s, next to the same layer on the fp32 MFMA.
//   hipcc -O3 --offload-arch=gfx950 tools/split_probe.hip -o tools/_split_probe && tools/_split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define DEVINL __device__ __forceinline__

// element j of half h at k-step s <-> accumulator register 8 (s & 1) + j of block s >> 1 (the bf16 kernels' correspondence)
DEVINL void split3(bf16x8 (&hi)[8], bf16x8 (&mid)[8], bf16x8 (&lo)[8], const f32x16 (&x)[4]) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[s >> 1][8 * (s & 1) + j];
            const __bf16 a = (__bf16)v;
            const float r1 = v - (float)a;
            const __bf16 b = (__bf16)r1;
            const float r2 = r1 - (float)b;
            hi[s][j] = a;
            mid[s][j] = b;
            lo[s][j] = (__bf16)r2;
        }
}
// w: [piece 0..2][s][t][lane] fragments of 8 bf16
DEVINL void layer_split(f32x16 (&acc)[4], const bf16x8 (&hi)[8], const bf16x8 (&mid)[8], const bf16x8 (&lo)[8], const bf16x8* p1, const bf16x8* p2,
                        const bf16x8* p3, int lane) {
    const bf16x8* w1 = p1 + lane;
    const bf16x8* w2 = p2 + lane;
    const bf16x8* w3 = p3 + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 a1 = w1[(s * 4 + t) * 64], a2 = w2[(s * 4 + t) * 64], a3 = w3[(s * 4 + t) * 64];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, hi[s], acc[t], 0, 0, 0);      // small terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, mid[s], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, lo[s], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, hi[s], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, mid[s], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, hi[s], acc[t], 0, 0, 0);
        }
}
// mode 0: split + six terms; 1: six terms without the split (bounds the VALU share); 2: fp32 MFMA reference layer (weights from LDS)
template <int MODE>
__global__ __launch_bounds__(512, 1) void k_layer(const uint16_t* wsplit, const float* wf32, float* out, int tiles_per_wave, float* check) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (MODE != 2) {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(wsplit);
        f32x4* d4 = reinterpret_cast<f32x4*>(smem);
        for (int i = threadIdx.x; i < 3 * 32768 / 16; i += blockDim.x) d4[i] = s4[i];
    } else {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(wf32);
        f32x4* d4 = reinterpret_cast<f32x4*>(smem);
        for (int i = threadIdx.x; i < 65536 / 16; i += blockDim.x) d4[i] = s4[i];
    }
    __syncthreads();
    f32x16 x[4], acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) x[t][k] = 0.01f * (float)((lane * 7 + t * 16 + k + wave + blockIdx.x) % 97) - 0.4f;
    bf16x8 hi[8], mid[8], lo[8];
    split3(hi, mid, lo, x);
    const int lane_in = lane;
    for (int it = 0; it < tiles_per_wave; ++it) {
        int lane = lane_in;
        asm volatile("" : "+v"(lane));      // keeps the (loop-invariant) LDS weight reads inside the loop: hoisted, they spill
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
        const bf16x8* l1 = reinterpret_cast<const bf16x8*>(smem);
        const bf16x8* g1 = reinterpret_cast<const bf16x8*>(wsplit);
        if (MODE == 0) {
            split3(hi, mid, lo, x);
            layer_split(acc, hi, mid, lo, l1, l1 + 2048, l1 + 4096, lane);
        } else if (MODE == 1) {
            layer_split(acc, hi, mid, lo, l1, l1 + 2048, l1 + 4096, lane);
        } else if (MODE == 3) {             // the low piece of the weights streamed from L2 (a third of the fragments, a sixth of the MFMAs)
            split3(hi, mid, lo, x);
            layer_split(acc, hi, mid, lo, l1, l1 + 2048, g1 + 4096, lane);
        } else if (MODE == 4) {             // low and middle pieces from L2
            split3(hi, mid, lo, x);
            layer_split(acc, hi, mid, lo, l1, g1 + 2048, g1 + 4096, lane);
        } else {
            const f32x4* wv = reinterpret_cast<const f32x4*>(smem) + lane;      // fragment order of the fp32 kernels: [j][lane][4]
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const f32x4 a = wv[j * 64];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], x[j >> 4][j & 15], acc[t], 0, 0, 0);
            }
        }
        // the next tile's input depends on this tile's output (as consecutive layers do), kept bounded
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) x[t][k] = fmaxf(acc[t][k], 0.f) * 0.05f + 0.01f * k;
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) s += acc[t][k];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (check && blockIdx.x == 0 && wave == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) check[(t * 16 + k) * 64 + lane] = acc[t][k];
    }
}

static uint16_t bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
    return (uint16_t)u;
}
static float bf16_val(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
int main() {
    const int L = 128, tiles = 400;
    std::vector<float> W((size_t)L * L);
    srand(1);
    for (auto& w : W) w = ((rand() % 20001) / 10000.0f - 1.0f) * sqrtf(6.0f / (2 * L));
    // split pieces in the bf16 kernels' fragment order: frag[(s * 4 + t) * 64 + lane][j] = W[k = f(s, h, j)][n = 32 t + c]
    std::vector<uint16_t> ws((size_t)3 * L * L);
    for (int s = 0; s < 8; ++s)
        for (int t = 0; t < 4; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int c = lane & 31, h = lane >> 5;
                    const int k = 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
                    const float w = W[(size_t)k * L + 32 * t + c];
                    const uint16_t a = bf16_bits(w);
                    const float r1 = w - bf16_val(a);
                    const uint16_t b = bf16_bits(r1);
                    const float r2 = r1 - bf16_val(b);
                    const size_t o = ((size_t)(s * 4 + t) * 64 + lane) * 8 + j;
                    ws[o] = a;
                    ws[(size_t)L * L + o] = b;
                    ws[(size_t)2 * L * L + o] = bf16_bits(r2);
                }
    uint16_t* dws;
    float *dwf, *dout;
    hipMalloc(&dws, ws.size() * 2);
    hipMalloc(&dwf, W.size() * 4);
    hipMalloc(&dout, 256 * 512 * 4);
    hipMemcpy(dws, ws.data(), ws.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dwf, W.data(), W.size() * 4, hipMemcpyHostToDevice);   // (the fp32 reference layer only needs SOME resident weights)
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_layer<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const char* names[5] = {"bf16 x 3 split, six terms (split per tile)", "six terms, no split (VALU share)", "fp32 MFMA 32x32x2 (reference layer)",
                            "split, low weight piece streamed from L2", "split, low and middle weight pieces from L2"};
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_layer<0>, dim3(256), dim3(512), 98304, 0, dws, dwf, dout, tiles, nullptr);
            if (mode == 1) hipLaunchKernelGGL(k_layer<1>, dim3(256), dim3(512), 98304, 0, dws, dwf, dout, tiles, nullptr);
            if (mode == 2) hipLaunchKernelGGL(k_layer<2>, dim3(256), dim3(512), 98304, 0, dws, dwf, dout, tiles, nullptr);
            if (mode == 3) hipLaunchKernelGGL(k_layer<3>, dim3(256), dim3(512), 98304, 0, dws, dwf, dout, tiles, nullptr);
            if (mode == 4) hipLaunchKernelGGL(k_layer<4>, dim3(256), dim3(512), 98304, 0, dws, dwf, dout, tiles, nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double flop = 2.0 * L * L * 32.0 * tiles * 8 * 256;      // fp32-equivalent flops of the layer
        printf("%-46s %.3f ms  = %.1f fp32-equivalent TFLOP/s\n", names[mode], best, flop / best * 1e-9);
    }
    return 0;
}
