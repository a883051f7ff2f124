// Is v_dot2(c)_f32_bf16 with a (1, 0) / (0, 1) selector an exact "convert one bf16 of a pair and add" (float(x) + c, one
// rounding)?  The bf16 processor kernels use it to unpack-and-accumulate in one VALU instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/dot2_probe.hip -o tools/_dot2_probe && tools/_dot2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* x, const float* c, float* lo, float* hi, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the selectors go through an opaque register: as compile-time constants hipcc (ROCm 7.2) folds the pair (1.0, 0.0) into the
    // inline constant `1.0`, which the instruction reads as the 32-bit pattern 0x3F800000 = the pair (0.0, 1.0) -- a wrong select
    unsigned u0 = 0x00003F80u, u1 = 0x3F800000u;
    asm volatile("" : "+s"(u0), "+s"(u1));
    const bf16x2 s0 = __builtin_bit_cast(bf16x2, u0), s1 = __builtin_bit_cast(bf16x2, u1);
    lo[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x[i]), s0, c[i], false);
    hi[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x[i]), s1, c[i], false);
}
int main() {
    const int n = 1 << 22;
    std::vector<unsigned> x(n);
    std::vector<float> c(n), lo(n), hi(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        // finite, normal bf16 values of every magnitude the latents see; c likewise (and exact zeros)
        auto rb = [] { unsigned e = 100 + rand() % 56, m = rand() & 0x7F, s = rand() & 1; return (s << 15) | (e << 7) | m; };
        x[i] = rb() | (rb() << 16);
        unsigned cb = ((unsigned)(rand() & 1) << 31) | ((unsigned)(100 + rand() % 56) << 23) | ((unsigned)rand() & 0x7FFFFF);
        if (i % 17 == 0) cb = 0;
        memcpy(&c[i], &cb, 4);
    }
    unsigned* dx; float *dc, *dl, *dh;
    hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dl, n * 4); hipMalloc(&dh, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dc, dl, dh, n);
    hipMemcpy(lo.data(), dl, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hi.data(), dh, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, maxulp = 0;
    for (int i = 0; i < n; ++i) {
        unsigned a = x[i] << 16, b = x[i] & 0xFFFF0000u;
        float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
        const float wl = fa + c[i], wh = fb + c[i];
        int il, iw, jh, jw;
        memcpy(&il, &lo[i], 4); memcpy(&iw, &wl, 4); memcpy(&jh, &hi[i], 4); memcpy(&jw, &wh, 4);
        const long d = std::max(labs((long)il - iw), labs((long)jh - jw));
        if (d > maxulp && d < (1L << 30)) maxulp = d;
        if (d > 1) {
            if (bad < 5) printf("mismatch x=%08x c=%g: lo %g want %g, hi %g want %g\n", x[i], c[i], lo[i], wl, hi[i], wh);
            ++bad;
        }
    }
    printf("dot2 probe: %ld results off by MORE than 1 fp32 ulp of %d; largest difference to the IEEE sum: %ld ulp\n", bad, n, maxulp);
    return bad ? 1 : 0;
}
