// Micro-benchmark (diagnostic, not product code): how busy can ONE wave per SIMD keep the fp32 MFMA pipe
// (v_mfma_f32_32x32x2_f32, 4 rotating accumulators, A operand from LDS) when independent VALU / VMEM work is interleaved
// into its instruction stream, compared with two waves per SIMD running MFMA chains only?
//   build:  hipcc --offload-arch=gfx950 -O3 -o tools/_mfma_probe tools/mfma_probe.hip      run: tools/_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VALU_PER_MFMA, bool WITH_VMEM>
__global__ void probe(float* out, const float* src, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1.0f + (float)(i & 7) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    float b[16], v[16];
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    for (int k = 0; k < 16; ++k) { b[k] = 1e-3f * (float)(lane + k); v[k] = (float)k; }
    const f32x4* w4 = reinterpret_cast<const f32x4*>(lds) + lane;
    const float* gp = src + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
    f32x4 g = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const f32x4 a = w4[j * 64];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[j & 15], acc[t], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < VALU_PER_MFMA; ++u) v[(4 * j + t + 5 * u) & 15] = v[(4 * j + t + 5 * u) & 15] * 1.0001f + 0.5f;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (WITH_VMEM && (j & 3) == 0) {          // one 16-byte load and one store per 16 MFMAs
                const f32x4 ld = *reinterpret_cast<const f32x4*>(gp + (size_t)((it * 16 + (j >> 2)) & 1023) * 65536);
                g += ld;
                *reinterpret_cast<f32x4*>(out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4 + (size_t)((j >> 2) & 15) * 1048576) = g;
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 16; ++k) s += acc[t][k];
    for (int k = 0; k < 16; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + g[0];
}

template <typename K>
static void run(const char* name, K kern, int threads, float* out, const float* src) {
    const int iters = 200, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 65536, 0, out, src, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 65536, 0, out, src, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * (threads / 64) * iters * 256.0 * 4096.0;
    printf("%-44s %7.3f ms  %6.1f TFLOP/s  (%.0f %% of 157.3)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *out, *src;
    hipMalloc(&out, (size_t)64 << 20);
    hipMalloc(&src, (size_t)1 << 30);
    hipMemset(src, 0, (size_t)1 << 30);
    run("1 wave/SIMD, MFMA only", probe<0, false>, 256, out, src);
    run("2 waves/SIMD, MFMA only", probe<0, false>, 512, out, src);
    run("1 wave/SIMD, MFMA + 3 VALU each", probe<3, false>, 256, out, src);
    run("1 wave/SIMD, MFMA + 6 VALU each", probe<6, false>, 256, out, src);
    run("1 wave/SIMD, MFMA + 10 VALU each", probe<10, false>, 256, out, src);
    run("1 wave/SIMD, MFMA + 3 VALU + VMEM", probe<3, true>, 256, out, src);
    run("2 waves/SIMD, MFMA + 3 VALU each", probe<3, false>, 512, out, src);
    run("3 waves/SIMD, MFMA + 3 VALU each", probe<3, false>, 768, out, src);
    run("4 waves/SIMD, MFMA + 3 VALU each", probe<3, false>, 1024, out, src);
    run("3 waves/SIMD, MFMA + 6 VALU each", probe<6, false>, 768, out, src);
    run("2 waves/SIMD, MFMA + 3 VALU + VMEM", probe<3, true>, 512, out, src);
    run("3 waves/SIMD, MFMA + 3 VALU + VMEM", probe<3, true>, 768, out, src);
    return 0;
}
