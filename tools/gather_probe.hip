// What a 16-byte-per-lane global load costs the CU's vector-memory path on gfx950, by access shape (design input: the edge kernels
// gather P / Q rows with lane = row; docs/experiments.md round 3 attributes ~8 % of k_edge_ring to that).  One block per CU, NW waves,
// every wave issues CH chains of 16 independent loads of one shape from an L2-resident table (rows of 512 B), waits, repeats.
//   shape 0  coalesced: the 64 lanes read 1 KiB contiguous (a tile-major fragment)                 8 lines per instruction
//   shape 1  row gather: lane (c, h) reads 16 B at row[idx[c]] + 32 m + 16 h (the P / Q gather)   32 lines per instruction
//   shape 2  the same with eight distinct rows per instruction (receiver-sorted Q rows)            8 lines
//   shape 3  64 distinct lines per instruction (rows 256 B apart per lane)                        64 lines
//   shape 4  row-contiguous: 32 lanes read one row's 512 B (two rows per instruction)              8 lines, full use
// Prints cycles per instruction and wave (s_memtime) and the implied bytes per clock and CU.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o tools/_gather_probe && tools/_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void k_probe(const f32x4* tab, const int* idx, int nrows, int reps, float* sink, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, t1 = 0;
    for (int rep = -1; rep < reps; ++rep) {
        if (rep == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        const int base = (int)(((unsigned)(blockIdx.x * 97 + wave * 131 + rep * 61) * 32u) % (unsigned)(nrows - 64));
        int row;
        if (SHAPE == 1) row = idx[(base + c) % nrows];
        else if (SHAPE == 2) row = idx[(base + (c >> 2)) % nrows];
        else row = base + c;
        f32x4 v[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const f32x4* p;
            if (SHAPE == 0) p = tab + (size_t)base * 32 + m * 64 + lane;                       // 1 KiB contiguous per instruction
            else if (SHAPE == 1 || SHAPE == 2) p = tab + (size_t)row * 32 + 2 * m + h;         // row-major rows, lane = row
            else if (SHAPE == 3) p = tab + ((size_t)(base + lane) * 32 + 2 * m) % ((size_t)nrows * 32);   // one line per lane
            else p = tab + (size_t)(base + 2 * m + h) * 32 + c;                                // 32 lanes per row
            v[m] = *p;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) acc += v[m];
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int SHAPE>
static void run(const char* name, int nw, const f32x4* tab, const int* idx, int nrows, float* sink, unsigned long long* cyc, int ncu) {
    const int reps = 400;
    hipLaunchKernelGGL(k_probe<SHAPE>, dim3(ncu), dim3(nw * 64), 0, 0, tab, idx, nrows, 20, sink, cyc);
    hipLaunchKernelGGL(k_probe<SHAPE>, dim3(ncu), dim3(nw * 64), 0, 0, tab, idx, nrows, reps, sink, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)ncu * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < ncu; ++b)
        for (int w = 0; w < nw; ++w) s += (double)h[(size_t)b * 8 + w];
    const double per_wave = s / (ncu * nw) / (reps * 16.0);          // cycles per instruction as one wave sees it
    const double per_cu = per_wave / nw;                              // CU-level cycles per instruction
    printf("  %-34s %d waves/CU: %7.1f cycles per instruction and wave, %6.1f per instruction on the CU, %6.1f B/clk/CU\n", name, nw, per_wave, per_cu,
           1024.0 / per_cu);
}

int main() {
    const int nrows = 8192;                       // 4 MiB table: L2-resident on every XCD
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    f32x4* tab; int* idx; float* sink; unsigned long long* cyc;
    hipMalloc(&tab, (size_t)nrows * 512);
    hipMalloc(&idx, nrows * 4);
    hipMalloc(&sink, 64);
    hipMalloc(&cyc, (size_t)ncu * 8 * 8);
    hipMemset(tab, 0, (size_t)nrows * 512);
    std::vector<int> hi(nrows);
    srand(1);
    for (int i = 0; i < nrows; ++i) hi[i] = rand() % nrows;
    hipMemcpy(idx, hi.data(), nrows * 4, hipMemcpyHostToDevice);
    printf("%s, %d CUs; 16-byte loads per lane from an L2-resident table\n", prop.gcnArchName, ncu);
    for (int nw : {1, 4, 8}) {
        run<0>("coalesced (8 lines)", nw, tab, idx, nrows, sink, cyc, ncu);
        run<4>("row-contiguous (8 lines)", nw, tab, idx, nrows, sink, cyc, ncu);
        run<2>("lane = row, 8 distinct rows", nw, tab, idx, nrows, sink, cyc, ncu);
        run<1>("lane = row, 32 distinct rows", nw, tab, idx, nrows, sink, cyc, ncu);
        run<3>("64 distinct lines", nw, tab, idx, nrows, sink, cyc, ncu);
    }
    return 0;
}
