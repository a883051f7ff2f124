// Diagnostic: what a kernel boundary costs inside a hipGraph replay (and eagerly) on this box: 31 dependent launches of a
// kernel that does nothing / that loads and stores one 16-byte value per thread, grids of 374 x 256 like the small-mesh pass.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(float* p) {}
__global__ void k_touch(float* p) {
    float4* q = reinterpret_cast<float4*>(p) + blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = *q;
    v.x += 1.f;
    *q = v;
}
template <typename K>
static void run(const char* name, K kern, float* buf) {
    hipStream_t st;
    hipStreamCreate(&st);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 31; ++i) hipLaunchKernelGGL(kern, dim3(374), dim3(256), 0, st, buf);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    const double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (200.0 * 31);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200 * 31; ++i) hipLaunchKernelGGL(kern, dim3(374), dim3(256), 0, st, buf);
    hipStreamSynchronize(st);
    const double us_eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (200.0 * 31);
    printf("%-10s per dependent launch: graph replay %.2f us, eager %.2f us\n", name, us_graph, us_eager);
}
int main() {
    float* buf;
    hipMalloc(&buf, 374 * 256 * 16);
    hipMemset(buf, 0, 374 * 256 * 16);
    run("empty", k_empty, buf);
    run("touch", k_touch, buf);
    return 0;
}
