// Diagnostic: what a grid-wide barrier costs inside ONE launch on this part -- the alternative to a kernel boundary (tools/launch_probe.hip:
// 1.8-2.2 us per dependent launch in graph replay) for a persistent small-mesh processor pass.  256 blocks x 256 threads (one per CU,
// co-resident on an idle device), 31 barriers per launch, with and without a 16-byte load + store per thread between barriers (as in
// launch_probe's k_touch; the store must be visible to the other XCDs' blocks behind the barrier: agent-scope release / acquire).
// A spin limit ends the wait (and flags the run) instead of hanging the device if the blocks were not co-resident.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* bad) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 22)) { *bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return true;
}
template <bool TOUCH>
__global__ __launch_bounds__(256) void k_barriers(float* p, unsigned* ctr, unsigned* bad, int nbar, unsigned base) {
    float4* q = reinterpret_cast<float4*>(p) + blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < nbar; ++i) {
        if (TOUCH) {
            // read what ANOTHER block wrote before the last barrier (the next block's slot), write our own
            const float4* o = reinterpret_cast<const float4*>(p) + ((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x;
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 w = __builtin_nontemporal_load(reinterpret_cast<const f4*>(o));
            float4 v = make_float4(w[0], w[1], w[2], w[3]);
            v.x += 1.f;
            *q = v;
        }
        grid_barrier(ctr, base + (unsigned)(i + 1) * gridDim.x, bad);
    }
}
template <bool TOUCH>
static void run(const char* name, float* buf, unsigned* ctr, unsigned* bad, int blocks) {
    hipStream_t st;
    (void)hipStreamCreate(&st);
    const int nbar = 31, reps = 200;
    unsigned base = 0;
    hipMemsetAsync(ctr, 0, 4, st);
    for (int i = 0; i < 5; ++i) { hipLaunchKernelGGL(k_barriers<TOUCH>, dim3(blocks), dim3(256), 0, st, buf, ctr, bad, nbar, base); base += (unsigned)nbar * blocks; }
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_barriers<TOUCH>, dim3(blocks), dim3(256), 0, st, buf, ctr, bad, nbar, base); base += (unsigned)nbar * blocks; }
    hipStreamSynchronize(st);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_barriers<TOUCH>, dim3(blocks), dim3(256), 0, st, buf, ctr, bad, 1, base); base += (unsigned)blocks; }
    hipStreamSynchronize(st);
    const double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    unsigned hb = 0;
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("%-6s %3d blocks: %.2f us per launch of %d barriers, %.2f us of 1 => %.2f us per barrier%s\n", name, blocks, us, nbar, us1, (us - us1) / (nbar - 1),
           hb ? "  (SPIN LIMIT HIT: blocks not co-resident?)" : "");
}
int main() {
    float* buf;
    unsigned *ctr, *bad;
    hipMalloc(&buf, 512 * 256 * 16);
    hipMemset(buf, 0, 512 * 256 * 16);
    hipMalloc(&ctr, 4);
    hipMalloc(&bad, 4);
    hipMemset(bad, 0, 4);
    for (int blocks : {128, 256}) {
        run<false>("empty", buf, ctr, bad, blocks);
        run<true>("touch", buf, ctr, bad, blocks);
    }
    return 0;
}
