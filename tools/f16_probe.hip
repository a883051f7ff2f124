// What v_mfma_f32_32x32x16_f16 does with subnormal fp16 operands, whether its products are exact in fp32, and its rate beside the bf16
// shape's (the question behind the two-piece fp16 split of the split path; docs/experiments.md, round 5).
//   hipcc --offload-arch=gfx950 -O3 tools/f16_probe.hip -o tools/_f16_probe && tools/_f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_denorm(const _Float16* a, const _Float16* b, float* out) {
    // A[m][k] = a[k] for every m, B[k][n] = b[k] for every n: every output element is sum_k a[k] b[k] (k = 0..15)
    const int lane = threadIdx.x;
    h8 A, B;
    for (int i = 0; i < 8; i++) { A[i] = a[(lane >> 5) * 8 + i]; B[i] = b[(lane >> 5) * 8 + i]; }
    f16v acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
    f16v acc[4] = {};
    h8 ah, bh; b8 ab, bb;
    for (int i = 0; i < 8; i++) { ah[i] = (_Float16)(threadIdx.x * 0.001f + i); bh[i] = (_Float16)(1.0f + i); ab[i] = (__bf16)(threadIdx.x * 0.001f + i); bb[i] = (__bf16)(1.0f + i); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
        }
    }
    float s = 0;
    for (int t = 0; t < 4; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_cvt(const float* x, float* out) {
    // the split as the kernels would do it: hi = RN16(x * s), r = x * s - hi (exact in fp32), lo = RN16(r)
    const float v = x[threadIdx.x], s = x[64];
    _Float16 hi = (_Float16)(v * s);
    float r = __builtin_fmaf(v, s, -(float)hi);
    _Float16 lo = (_Float16)r;
    out[threadIdx.x * 2] = (float)hi; out[threadIdx.x * 2 + 1] = (float)lo;
}

int main() {
    _Float16 ha[16], hb[16]; float* dout; _Float16 *da, *db;
    hipMalloc(&dout, 1 << 20); hipMalloc(&da, 32); hipMalloc(&db, 32);
    auto run = [&](const char* what, double expect) {
        hipMemcpy(da, ha, 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, 32, hipMemcpyHostToDevice);
        k_denorm<<<1, 64>>>(da, db, dout); float r; hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
        printf("%-58s got %.9e expect %.9e %s\n", what, r, expect, (double)r == expect ? "EXACT" : "DIFFERS");
    };
    for (int i = 0; i < 16; i++) { ha[i] = 0; hb[i] = 0; }
    ha[0] = (_Float16)ldexpf(1.0f, -20); hb[0] = (_Float16)1.0f;                 // subnormal A
    run("subnormal 2^-20 x 1", ldexp(1.0, -20));
    ha[0] = (_Float16)1.0f; hb[0] = (_Float16)ldexpf(3.0f, -24);                 // subnormal B (two low bits)
    run("1 x subnormal 3 * 2^-24", ldexp(3.0, -24));
    ha[0] = (_Float16)ldexpf(1.0f, -20); hb[0] = (_Float16)ldexpf(1.0f, -20);    // product 2^-40: fp32 normal
    run("subnormal x subnormal = 2^-40", ldexp(1.0, -40));
    ha[0] = (_Float16)(1.0f + ldexpf(1.0f, -10)); hb[0] = (_Float16)(1.0f + ldexpf(1.0f, -10));   // 22-bit product
    run("(1 + 2^-10)^2 exact in fp32", (1.0 + ldexp(1.0, -10)) * (1.0 + ldexp(1.0, -10)));
    for (int i = 0; i < 16; i++) { ha[i] = (_Float16)(1.0f + i * ldexpf(1.0f, -10)); hb[i] = (_Float16)(0.5f + i * ldexpf(1.0f, -9)); }
    { double e = 0; for (int i = 0; i < 16; i++) e += (double)(float)ha[i] * (double)(float)hb[i]; run("16-term dot product (one fp32 rounding at most)", (double)(float)e); }
    ha[0] = (_Float16)60000.0f; hb[0] = (_Float16)60000.0f; for (int i = 1; i < 16; i++) ha[i] = hb[i] = 0;
    run("60000 x 60000 (no overflow in fp32)", 3.6e9);
    // the split
    float hx[65] = {}; hx[0] = 0.1f; hx[1] = 1.2345678f; hx[2] = 3.0e-5f; hx[3] = -7.654321e-3f; hx[64] = 4096.0f;
    float* dx; hipMalloc(&dx, sizeof hx); hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(dx, dout); float hc[8]; hipMemcpy(hc, dout, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++) printf("split %.9e * 4096: hi %.9e lo %.9e  hi+lo-x*s %.3e (x*s ulp %.3e)\n", hx[i], hc[2 * i], hc[2 * i + 1],
                                       (double)hc[2 * i] + (double)hc[2 * i + 1] - (double)hx[i] * 4096.0, ldexp(1.0, ilogb(hx[i] * 4096.0) - 23));
    // rate
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; kind++) {
        const int iters = 4096, blocks = 256 * 8;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (kind == 0) k_rate<0><<<blocks, 256>>>(dout, iters); else k_rate<1><<<blocks, 256>>>(dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 4 * 32768.0;
        printf("%s: %.1f TFLOP/s\n", kind == 0 ? "v_mfma_f32_32x32x16_f16 " : "v_mfma_f32_32x32x16_bf16", flop / ms * 1e-9);
    }
    return 0;
}
