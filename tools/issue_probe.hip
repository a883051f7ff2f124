// Does a wave issuing back-to-back MFMAs keep the OTHER wave of its SIMD from issuing VALU instructions?
// One block of 8 waves per CU (2 per SIMD): waves 0-3 run a chain of independent v_mfma_f32_32x32x2_f32 (optionally with
// `s_nop K` behind each), waves 4-7 run a loop of independent v_fma_f32.  Each role is timed alone and together.
//   hipcc -O3 --offload-arch=gfx950 tools/issue_probe.hip -o tools/_issue_probe && tools/_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NOP>
__device__ __forceinline__ void mfma_role(int iters, float* sink, int lane) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = 1.0f + lane * 1e-3f, y = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
            if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP >= 0 ? NOP : 0));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
            if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP >= 0 ? NOP : 0));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
            if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP >= 0 ? NOP : 0));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
            if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP >= 0 ? NOP : 0));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    sink[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NOP>
__device__ __forceinline__ void mfma16_role(int iters, float* sink, int lane) {   // 6 independent 16x16x4 accumulators
    f32x4 a[6] = {};
    float x = 1.0f + lane * 1e-3f, y = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(x), "v"(y));
                if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP >= 0 ? NOP : 0));
            }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    sink[threadIdx.x] = a[0][0] + a[1][1] + a[2][2] + a[3][3] + a[4][0] + a[5][1];
}
template <int NOP>
__global__ __launch_bounds__(256, 1) void k_probe16(int mi, float* sink, unsigned long long* t) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    mfma16_role<NOP>(mi, sink + blockIdx.x * 256, lane);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) t[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int NOP>
void run16(const char* name, int mi) {
    float* sink;
    unsigned long long* t;
    hipMalloc(&sink, 256 * 256 * 4);
    hipMalloc(&t, 256 * 4 * 8);
    unsigned long long h[256 * 4];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_probe16<NOP>, dim3(256), dim3(256), 0, 0, mi, sink, t);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256 * 4; ++i) m += h[i] * 0.01 / (256 * 4);
    printf("16x16x4 %-8s %.1f us for %d MFMAs per wave = %.1f clocks each at 2.4 GHz (%.1f TFLOP/s on 1024 SIMDs)\n", name, m, 24 * mi,
           m * 2400.0 / (24 * mi), 24.0 * mi * 2048 * 1024 / m * 1e-6);
    hipFree(sink);
    hipFree(t);
}
// bf16 matrix instruction (v_mfma_f32_32x32x16_bf16: 8 passes... measured below) beside VALU: do THEY overlap?
template <int NOP>
__device__ __forceinline__ void mfma_bf16_role(int iters, float* sink, int lane) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f32x4 x = {1.0f + lane * 1e-3f, 0.5f, 0.25f, 2.0f}, y = {0.5f, 1.5f, 0.75f, 1.0f};   // bit patterns only
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
            if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP >= 0 ? NOP : 0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
            if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP >= 0 ? NOP : 0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
            if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP >= 0 ? NOP : 0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
            if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP >= 0 ? NOP : 0));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    sink[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__device__ __forceinline__ void valu_role(int iters, float* sink, int lane);
template <int NOP>
__global__ __launch_bounds__(512, 1) void k_probe_bf(int mode, int mi, int vi, float* sink, unsigned long long* t) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (wave < 4) {
        if (mode & 1) mfma_bf16_role<NOP>(mi, sink + blockIdx.x * 512, lane);
    } else {
        if (mode & 2) valu_role(vi, sink + blockIdx.x * 512, lane);
    }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) t[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int NOP>
void run_bf(const char* name, int mi, int vi) {
    float* sink;
    unsigned long long* t;
    hipMalloc(&sink, 256 * 512 * 4);
    hipMalloc(&t, 256 * 8 * 8);
    unsigned long long h[256 * 8];
    double res[4][2];
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_probe_bf<NOP>, dim3(256), dim3(512), 0, 0, mode, mi, vi, sink, t);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[b * 8 + w] * 0.01 / (256 * 4);
        res[mode][0] = m;
        res[mode][1] = v;
    }
    const double mf = 16.0 * mi * 32 * 32 * 16 * 2;
    printf("bf16 %-8s MFMA alone %.1f us (%.0f TFLOP/s on 1024 SIMDs, %.1f clocks each at 2.4 GHz) | VALU alone %.1f us | together: MFMA %.1f us, VALU %.1f us\n",
           name, res[1][0], mf * 1024 / res[1][0] * 1e-6, res[1][0] * 2400.0 / (16.0 * mi), res[2][1], res[3][0], res[3][1]);
    hipFree(sink);
    hipFree(t);
}
__device__ __forceinline__ void valu_role(int iters, float* sink, int lane) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = lane * 0.01f + k;
    const float m = 1.0001f, c = 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(c));
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    sink[threadIdx.x] = s;
}
// mode bit 0: MFMA waves active, bit 1: VALU waves active
template <int NOP>
__global__ __launch_bounds__(512, 1) void k_probe(int mode, int mi, int vi, float* sink, unsigned long long* t) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (wave < 4) {
        if (mode & 1) mfma_role<NOP>(mi, sink + blockIdx.x * 512, lane);
    } else {
        if (mode & 2) valu_role(vi, sink + blockIdx.x * 512, lane);
    }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) t[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int NOP>
void run(const char* name, int mi, int vi) {
    float* sink;
    unsigned long long* t;
    hipMalloc(&sink, 256 * 512 * 4);
    hipMalloc(&t, 256 * 8 * 8);
    unsigned long long h[256 * 8];
    double res[4][2];
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_probe<NOP>, dim3(256), dim3(512), 0, 0, mode, mi, vi, sink, t);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[b * 8 + w] * 0.01 / (256 * 4);
        res[mode][0] = m;
        res[mode][1] = v;
    }
    const double mf = 16.0 * mi * 32 * 32 * 2 * 2;   // flops per MFMA wave
    printf("%-10s MFMA alone %.1f us (%.1f TFLOP/s on 1024 SIMDs) | VALU alone %.1f us | together: MFMA %.1f us, VALU %.1f us\n", name, res[1][0],
           mf * 1024 / res[1][0] * 1e-6, res[2][1], res[3][0], res[3][1]);
    hipFree(sink);
    hipFree(t);
}
int main() {
    const int mi = 2000, vi = 4000;   // 32 k MFMAs (~0.9 ms) ; 256 k VALU (~0.5 ms alone)
    run<-1>("no nop", mi, vi);
    run<0>("s_nop 0", mi, vi);
    run<3>("s_nop 3", mi, vi);
    run<7>("s_nop 7", mi, vi);
    run<11>("s_nop 11", mi, vi);
    run<13>("s_nop 13", mi, vi);
    run<15>("s_nop 15", mi, vi);
    run_bf<-1>("no nop", 4000, vi);
    run_bf<1>("s_nop 1", 4000, vi);
    run_bf<3>("s_nop 3", 4000, vi);
    run16<-1>("no nop", 1500);
    run16<0>("s_nop 0", 1500);
    run16<1>("s_nop 1", 1500);
    run16<3>("s_nop 3", 1500);
    run16<5>("s_nop 5", 1500);
    return 0;
}
