// Micro-benchmark (diagnostic, not product code): what the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) sustains on this part, and how
// much VALU work hides in its shadow depending on WHERE in a wave's stream it sits.  Operands live in registers (no LDS, no memory
// in the loop), four independent accumulators, 256 blocks.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_mfma_probe tools/mfma_probe.hip && tools/_mfma_probe
// Round-1's version read the A operand from LDS inside the loop and topped out at 138.7 TFLOP/s; that was the LDS read, not the
// pipe: with register operands the pipe delivers the guide's ~155 TFLOP/s (99 % of 157.3).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMAs only.  MODE 1: V VALU instructions after EVERY MFMA.  MODE 2: 4 V VALU instructions after every group of 4 MFMAs
// (the same instruction counts as MODE 1, placed the way a compiler clusters them).
template <int V, int MODE>
__global__ void probe(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    float a[4], b[16], v[16];
    for (int t = 0; t < 4; ++t) {
        a[t] = 1.0f + 1e-3f * (float)(lane + t);
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    }
    for (int k = 0; k < 16; ++k) { b[k] = 1e-3f * (float)(lane + k); v[k] = (float)k; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[j & 15], acc[t], 0, 0, 0);
                if (MODE == 1) {
#pragma unroll
                    for (int u = 0; u < V; ++u) v[(4 * j + t + 5 * u) & 15] = v[(4 * j + t + 5 * u) & 15] * 1.0001f + 0.5f;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 2) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4 * V; ++u) v[(j + 5 * u) & 15] = v[(j + 5 * u) & 15] * 1.0001f + 0.5f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 16; ++k) s += acc[t][k];
    for (int k = 0; k < 16; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int threads, float* out) {
    const int iters = 400, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * (threads / 64) * iters * 256.0 * 4096.0;
    printf("%-64s %7.3f ms  %6.1f TFLOP/s  (%.0f %% of 157.3)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float* out;
    (void)hipMalloc(&out, (size_t)64 << 20);
    run("1 wave/SIMD, MFMA only", probe<0, 0>, 256, out);
    run("2 waves/SIMD, MFMA only", probe<0, 0>, 512, out);
    run("1 wave/SIMD, 3 VALU after EVERY MFMA", probe<3, 1>, 256, out);
    run("1 wave/SIMD, 6 VALU after EVERY MFMA", probe<6, 1>, 256, out);
    run("1 wave/SIMD, 10 VALU after EVERY MFMA", probe<10, 1>, 256, out);
    run("1 wave/SIMD, 14 VALU after EVERY MFMA", probe<14, 1>, 256, out);
    run("1 wave/SIMD, 12 VALU after every 4 MFMAs (= 3 per MFMA)", probe<3, 2>, 256, out);
    run("1 wave/SIMD, 24 VALU after every 4 MFMAs (= 6 per MFMA)", probe<6, 2>, 256, out);
    run("1 wave/SIMD, 40 VALU after every 4 MFMAs (= 10 per MFMA)", probe<10, 2>, 256, out);
    run("2 waves/SIMD, 3 VALU after EVERY MFMA", probe<3, 1>, 512, out);
    run("2 waves/SIMD, 12 VALU after every 4 MFMAs", probe<3, 2>, 512, out);
    run("2 waves/SIMD, 40 VALU after every 4 MFMAs", probe<10, 2>, 512, out);
    return 0;
}
