// Which waves of a 512-thread workgroup share a SIMD on gfx950?  (HW_REG_HW_ID: wave[3:0] simd[5:4] cu[11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d);
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 3; ++b) { printf("block %d simd of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b*8+w] >> 4) & 3); printf("   cu:"); for (int w = 0; w < 8; ++w) printf(" %u", (h[b*8+w] >> 8) & 15); printf("\n"); }
    int pair04 = 0, pair01 = 0;
    for (int b = 0; b < 256; ++b) { bool a = true, c = true; for (int w = 0; w < 4; ++w) a &= ((h[b*8+w]>>4)&3) == ((h[b*8+w+4]>>4)&3); for (int w = 0; w < 8; w += 2) c &= ((h[b*8+w]>>4)&3) == ((h[b*8+w+1]>>4)&3); pair04 += a; pair01 += c; }
    printf("blocks where waves (w,w+4) share a SIMD: %d/256; where (2k,2k+1) share: %d/256\n", pair04, pair01);
    return 0;
}
